"""Synthetic 8-bit 4:2:0 frames (SURVEY.md section 8d generators).

`smooth` is BASELINE config 1's 416x240 plumbing frame, `textured` the config 2-5 content.
Both are pure numpy so that the same seeded frame can be produced in the build container
and on the GPU box.
"""
import numpy as np


def smooth(width, height, seed=1234):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    Y = 128 + 40 * np.sin(x / 17) + 30 * np.cos(y / 11) + 20 * ((x // 32) % 2) + rng.normal(0, 6, (height, width))
    Y[60:120, 100:220] += 50
    cy, cx = np.mgrid[0:height // 2, 0:width // 2].astype(np.float64)
    U = 128 + 20 * np.sin(cx / 23) + rng.normal(0, 2, cy.shape)
    V = 128 + 20 * np.cos(cy / 19) + rng.normal(0, 2, cy.shape)
    f = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return f(Y), f(U), f(V)


def survey_frame(width=416, height=240, seed=1234):
    """The frame of the survey's reference run of BASELINE config 0 (416x240, QP 32, 8448 bits, PSNR
    32.3524 / 41.0897 / 41.1974 dB).  The survey describes it but did not keep the file; of the readings of
    that description, this one -- chroma waves on luma coordinates, float -> uint8 by truncation -- is the one on which
    decide -> deblock -> SAO reproduces all three PSNR values to the fourth decimal (tests/test_sao.py)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    Y = 128 + 40 * np.sin(x / 17) + 30 * np.cos(y / 11) + 20 * ((x // 32) % 2) + rng.normal(0, 6, (height, width))
    Y[60:120, 100:220] += 50
    cy, cx = np.mgrid[0:height // 2, 0:width // 2].astype(np.float64)
    U = 128 + 20 * np.sin(2 * cx / 23) + rng.normal(0, 2, cy.shape)
    V = 128 + 20 * np.cos(2 * cy / 19) + rng.normal(0, 2, cy.shape)
    f = lambda a: np.clip(a, 0, 255).astype(np.uint8)
    return f(Y), f(U), f(V)


def textured(width, height, seed=7):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    noise = rng.normal(0, 18, (height, width)) * (1 + ((x // 64 + y // 64) % 3)) / 2
    Y = 128 + 50 * np.sin(x / 9 + y / 31) + 35 * np.cos(y / 7) * np.sin(x / 53) + noise
    cy, cx = np.mgrid[0:height // 2, 0:width // 2].astype(np.float64)
    U = 128 + 25 * np.sin(cx / 23) + rng.normal(0, 5, cy.shape)
    V = 128 + 25 * np.cos(cy / 19) + rng.normal(0, 5, cy.shape)
    f = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return f(Y), f(U), f(V)


def mixed(width, height, seed=7):
    """Piecewise content (flat blocks of 4..64 px, ramps, sharp text-like edges, mild noise):
    drives decisions through all CU depths, NxN, RQT splits and transform skip."""
    rng = np.random.default_rng(seed)
    Y = np.zeros((height, width), np.float64)
    for bs in (64, 32, 16, 8, 4):
        gy, gx = (height + bs - 1) // bs, (width + bs - 1) // bs
        lvl = rng.integers(16, 235, (gy, gx)).astype(np.float64)
        mask = rng.random((gy, gx)) < (0.55 if bs == 64 else 0.3)
        up = np.kron(lvl, np.ones((bs, bs)))[:height, :width]
        m = np.kron(mask, np.ones((bs, bs)))[:height, :width] > 0
        if bs == 64:
            Y = up
        else:
            Y = np.where(m, up, Y)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    Y = Y + 12 * np.sin(x / 5.0) * (((x // 48) + (y // 40)) % 3 == 0) + rng.normal(0, 1.5, (height, width))
    cy, cx = np.mgrid[0:height // 2, 0:width // 2].astype(np.float64)
    U = 128 + 0.35 * (Y[::2, ::2] - 128) * (((cx // 24) % 2) * 2 - 1) + rng.normal(0, 1.0, cy.shape)
    V = 128 + 30 * np.sin(cx / 13 + cy / 29) + 40 * ((cy // 8 + cx // 8) % 5 == 0) + rng.normal(0, 1.0, cy.shape)
    f = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return f(Y), f(U), f(V)
