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
