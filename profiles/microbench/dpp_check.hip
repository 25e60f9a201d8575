// dpp_check.hip -- which way the gfx950 DPP row shifts move data, and the two reductions qm_wave.h builds from them.
// hipcc --offload-arch=gfx950 -O2 dpp_check.hip -o /tmp/dpp_check && /tmp/dpp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CTRL> __device__ __forceinline__ int dpp_maxT(int v) { int t = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); return t > v ? t : v; }
template <int CTRL> __device__ __forceinline__ int dpp_minT(int v) { int t = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); return t < v ? t : v; }
#define dpp_max(v, c) dpp_maxT<c>(v)
#define dpp_min(v, c) dpp_minT<c>(v)
__global__ void k(const int* in, int* out) {
  const int l = threadIdx.x;
  int v = in[l];
  out[l] = __builtin_amdgcn_update_dpp(-1, v, 0x101, 0xf, 0xf, false);          // row_shl:1
  out[64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false);     // row_shr:1
  // wave max: row_shr 1,2,4,8 then row_bcast15, row_bcast31 -> lane 63
  int m = v;
  m = dpp_max(m, 0x111); m = dpp_max(m, 0x112); m = dpp_max(m, 0x114); m = dpp_max(m, 0x118);
  { int t = __builtin_amdgcn_update_dpp(m, m, 0x142, 0xa, 0xf, false); m = t > m ? t : m; }
  { int t = __builtin_amdgcn_update_dpp(m, m, 0x143, 0xc, 0xf, false); m = t > m ? t : m; }
  out[128 + l] = m;
  // group min to the leader, G = 16: row_shl 1,2,4,8
  int g = v;
  g = dpp_min(g, 0x101); g = dpp_min(g, 0x102); g = dpp_min(g, 0x104); g = dpp_min(g, 0x108);
  out[192 + l] = g;
  // butterfly inside a row (every lane gets the minimum of its group of 4 / 8 / 16): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
  int b = v;
  b = dpp_min(b, 0xB1); b = dpp_min(b, 0x4E); out[256 + l] = b;
  b = dpp_min(b, 0x141); out[320 + l] = b;
  b = dpp_min(b, 0x140); out[384 + l] = b;
}
int main() {
  int h[64], o[448]; srand(7);
  for (int i = 0; i < 64; ++i) h[i] = rand() % 1000;
  int *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 1792);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout); hipMemcpy(o, dout, 1792, hipMemcpyDeviceToHost);
  int shl_ok = 1, shr_ok = 1;
  for (int i = 0; i < 64; ++i) { int e = (i % 16 != 15) ? h[i + 1] : -1; if (o[i] != e) shl_ok = 0; int f = (i % 16 != 0) ? h[i - 1] : -1; if (o[64 + i] != f) shr_ok = 0; }
  int mx = h[0]; for (int i = 1; i < 64; ++i) mx = h[i] > mx ? h[i] : mx;
  int gm_ok = 1; for (int r = 0; r < 4; ++r) { int mn = h[16 * r]; for (int i = 1; i < 16; ++i) mn = h[16 * r + i] < mn ? h[16 * r + i] : mn; if (o[192 + 16 * r] != mn) gm_ok = 0; }
  int bf_ok = 1;
  for (int G = 4, slot = 256; G <= 16; G *= 2, slot += 64)
    for (int i = 0; i < 64; ++i) { int mn = 1 << 30; for (int j = i / G * G; j < i / G * G + G; ++j) mn = h[j] < mn ? h[j] : mn; if (o[slot + i] != mn) bf_ok = 0; }
  printf("butterfly min over groups of 4 / 8 / 16 in every lane: %s\n", bf_ok ? "yes" : "NO");
  printf("row_shl:1 dst[i]=src[i+1]: %s   row_shr:1 dst[i]=src[i-1]: %s   wave max in lane 63: %s (%d vs %d)   group-of-16 min in lane 0 of the row: %s\n",
         shl_ok ? "yes" : "NO", shr_ok ? "yes" : "NO", o[128 + 63] == mx ? "yes" : "NO", o[128 + 63], mx, gm_ok ? "yes" : "NO");
  return 0;
}
