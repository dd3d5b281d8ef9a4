/*
 * dbk_emu.cpp -- TEST-ONLY: the deblocking kernel source (csrc/fcu_deblock.h) compiled for the CPU with the HIP
 * keywords defined away and the grid run as a loop, so that its indexing and arithmetic can be checked against the
 * oracle / the reference's golden vectors without a GPU.  Not part of libfcu.so.
 */
#define FCU_EMU 1
#include "../../fast-cu-decision-hevc_amd/csrc/fcu_host.h"
#define __device__
#define __global__
#define __launch_bounds__(x)
struct Dim3 { unsigned x, y, z; };
static thread_local Dim3 blockIdx, threadIdx;
#include "../../fast-cu-decision-hevc_amd/csrc/fcu_deblock.h"

using namespace fcu;

extern "C" void dbk_emu(const fcu_ctu_out *out, uint8_t *y, uint8_t *u, uint8_t *v, int w, int h, int boff, int toff)
{
  const int w_ctu = (w + 63) / 64;
  const unsigned n0 = (unsigned)((w >> 3) * (h >> 2)), n1 = (unsigned)((w >> 2) * (h >> 3));
  for (unsigned b = 0; b < (n0 + DBK_THREADS - 1) / DBK_THREADS; b++)
    for (unsigned t = 0; t < DBK_THREADS; t++) { blockIdx.x = b; threadIdx.x = t; dbk_pass<0>(out, y, u, v, w, h, w_ctu, boff, toff); }
  for (unsigned b = 0; b < (n1 + DBK_THREADS - 1) / DBK_THREADS; b++)
    for (unsigned t = 0; t < DBK_THREADS; t++) { blockIdx.x = b; threadIdx.x = t; dbk_pass<1>(out, y, u, v, w, h, w_ctu, boff, toff); }
}
