// Which compute units does a hipExtStreamCreateWithCUMask stream really run on?  (LOG.md A.17, round 6: tools/a17_lab.py --mask)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/cu_census.hip -o tools/bin/cu_census && tools/bin/cu_census
// For each mask of tools/a17_lab.py (victim / aggressor side) 8192 blocks of 64 threads record HW_REG_XCC_ID and HW_REG_HW_ID
// (se_id, sh_id, cu_id); the summary is the number of distinct CUs seen per XCC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
#include <string>

__global__ void census_kernel(uint32_t* out, int spin) {
  uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));     // HW_REG_XCC_ID[3:0]
  uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));      // HW_REG_HW_ID
  volatile int sink = 0;
  for (int i = 0; i < spin; ++i) sink += i;                                       // stay resident long enough for the grid to spread
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char* name, const std::vector<int>& bits) {
  uint32_t words[8] = {0};
  for (int b : bits) words[b / 32] |= 1u << (b % 32);
  hipStream_t st;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, words);
  if (e != hipSuccess) { printf("%-28s hipExtStreamCreateWithCUMask -> %d\n", name, (int)e); return; }
  const int nb = 8192;
  uint32_t* d;
  hipMalloc(&d, nb * 8);
  hipLaunchKernelGGL(census_kernel, dim3(nb), dim3(64), 0, st, d, 20000);
  hipStreamSynchronize(st);
  std::vector<uint32_t> h(2 * nb);
  hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  std::map<int, std::set<uint32_t>> per;
  for (int b = 0; b < nb; ++b) {
    const uint32_t hw = h[2 * b + 1];
    const uint32_t cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per[h[2 * b] & 15].insert((se << 8) | (sh << 4) | cu);
  }
  printf("%-28s mask bits %3zu | CUs seen per XCC:", name, bits.size());
  int total = 0;
  for (int x = 0; x < 8; ++x) { printf(" %2zu", per.count(x) ? per[x].size() : (size_t)0); total += per.count(x) ? (int)per[x].size() : 0; }
  printf(" | total %d\n", total);
  if (bits.size() <= 32) {
    printf("%-28s   (xcc: se.sh.cu)", "");
    for (auto& kv : per) for (uint32_t c : kv.second) printf(" %d:%d.%d.%d", kv.first, c >> 8, (c >> 4) & 1, c & 15);
    printf("\n");
  }
  hipFree(d);
  hipStreamDestroy(st);
}

int main() {
  std::vector<int> all, lo32, hi224, xcd0, notxcd0;
  for (int i = 0; i < 256; ++i) {
    all.push_back(i);
    (i < 32 ? lo32 : hi224).push_back(i);
    (i % 8 == 0 ? xcd0 : notxcd0).push_back(i);
  }
  run("all", all);
  run("spread32 victim (bits 0..31)", lo32);
  run("spread32 aggr (bits 32..255)", hi224);
  run("xcd0 victim (i % 8 == 0)", xcd0);
  run("xcd0 aggr (i % 8 != 0)", notxcd0);
  run("bits 0..7", std::vector<int>{0, 1, 2, 3, 4, 5, 6, 7});
  run("bits 0,8,16,24", std::vector<int>{0, 8, 16, 24});
  return 0;
}
