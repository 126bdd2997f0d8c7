// diagnostic: which (XCC, SE, CU) do the workgroups of a stream created with hipExtStreamCreateWithCUMask land on?
// Calibrates the bit -> CU mapping that tools/cu_partition.py relies on: bit i of the mask is taken as CU slot (i / 8) of XCC (i % 8)
// (the KFD spreads successive bits over the XCCs first, then over the shader engines of an XCC).  Prints, for a few masks, the set of
// distinct (xcc, se, cu) triples seen by 16384 short workgroups.   hipcc --offload-arch=gfx950 cu_mask_map.hip -o cu_mask_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void where(unsigned* out, int spin) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID, bits 3:0
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
}
static int run(const char* name, const unsigned* mask, int words, unsigned* d, int nb) {
  hipStream_t st;
  if (mask) CK(hipExtStreamCreateWithCUMask(&st, words, mask)); else CK(hipStreamCreate(&st));
  hipLaunchKernelGGL(where, dim3(nb), dim3(256), 0, st, d, 40);
  CK(hipStreamSynchronize(st));
  std::vector<unsigned> h(nb);
  CK(hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost));
  std::set<unsigned> cus; int per_xcc[16] = {}; std::set<unsigned> se_cu;
  for (int b = 0; b < nb; ++b) {
    const unsigned xcc = h[b] >> 16, cu = (h[b] >> 8) & 15, sh = (h[b] >> 12) & 1, se = (h[b] >> 13) & 7;
    if (cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu).second) per_xcc[xcc & 15]++;
    se_cu.insert((se << 8) | (sh << 4) | cu);
  }
  // does workgroup b run on XCC b % 8 (the assumption of the XCD-aware tile orders)?
  int rr = 0;
  for (int b = 0; b < nb; ++b) rr += (int)((h[b] >> 16) & 15) == b % 8;
  printf("%-28s distinct CUs %3zu  per XCC:", name, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %2d", per_xcc[x]);
  printf("   blockIdx %% 8 == xcc for %5.1f %% of the workgroups   (se,sh,cu) slots used:", 100.0 * rr / nb);
  for (unsigned v : se_cu) printf(" %u.%u.%u", v >> 8, (v >> 4) & 1, v & 15);
  printf("\n");
  CK(hipStreamDestroy(st));
  return 0;
}
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s, %d CUs\n", pr.name, pr.multiProcessorCount);
  const int nb = 16384; unsigned* d; CK(hipMalloc(&d, nb * 4));
  if (run("no mask", nullptr, 0, d, nb)) return 1;
  unsigned m[8];
  auto fill = [&](int lo, int hi) { for (int w = 0; w < 8; ++w) m[w] = 0; for (int i = 0; i < 256; ++i) if (i / 8 >= lo && i / 8 < hi) m[i / 32] |= 1u << (i % 32); };
  fill(0, 32); if (run("all 256 bits", m, 8, d, nb)) return 1;
  fill(0, 24); if (run("slots 0..23 of every XCC", m, 8, d, nb)) return 1;
  fill(24, 32); if (run("slots 24..31 of every XCC", m, 8, d, nb)) return 1;
  fill(0, 16); if (run("slots 0..15 of every XCC", m, 8, d, nb)) return 1;
  fill(0, 1); if (run("slot 0 of every XCC", m, 8, d, nb)) return 1;
  for (int w = 0; w < 8; ++w) m[w] = 0;
  for (int i = 0; i < 256; ++i) if (i % 8 < 6) m[i / 32] |= 1u << (i % 32);
  printf("(not run: a mask that leaves whole XCCs without CUs -- bits with i %% 8 >= 6 cleared -- could strand the workgroups the\n"
         " dispatcher hands to those XCCs; the partition used by tools/cu_partition.py keeps CUs of EVERY XCC in both streams)\n");
  return 0;
}
