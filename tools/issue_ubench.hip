// Micro-benchmark: at what rate does ONE wave issue instructions on an MI355X, by kind and by
// encoding size?  The decode walk (fqcomp28_amd/csrc/decode.hip) is one wave per stream and spends
// ~8 cycles per instruction; this separates issue rate, dependency latency and instruction fetch.
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_ubench.hip -o /tmp/issue_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define BODY(txt) asm volatile(".rept 64\n\t" txt "\n\t.endr" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+v"(x), "+v"(y), "+v"(z) : : "scc", "vcc")
#define BODY32(txt) asm volatile(".rept 32\n\t" txt "\n\t.endr" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+v"(x), "+v"(y), "+v"(z) : : "scc", "vcc")
#define BODY16(txt) asm volatile(".rept 16\n\t" txt "\n\t.endr" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+v"(x), "+v"(y), "+v"(z) : : "scc", "vcc")

template <int VAR>
__global__ void __launch_bounds__(64) k(unsigned iters, unsigned *out, unsigned long long *cyc, const unsigned *tab) {
  __shared__ unsigned lds[1024];
  lds[threadIdx.x] = threadIdx.x * 4u & 255u;
  __syncthreads();
  const unsigned lds_at = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)lds;
  unsigned zero = 0;
  asm volatile("" : "+v"(zero));
  asm volatile("v_mov_b32 v13, 0\n\tv_mov_b32 v16, 1\n\tv_mov_b32 v17, 2\n\tv_mov_b32 v18, 3\n\tv_mov_b32 v19, 4" ::: "v13", "v16", "v17", "v18", "v19");
#define TWELVE "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
#define MEMBODY(txt) asm volatile(".rept 4\n\t" txt "\n\t.endr" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+v"(x), "+v"(y), "+v"(z) : "s"(tab), "s"(lds_at), "v"(zero) : "scc", "vcc", "memory", "v10", "v11", "v12", "v13", "v16", "v17", "v18", "v19")
  unsigned a = 1, b = 2, c = 3, d = 4, x = threadIdx.x, y = 7, z = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned i = 0; i < iters; i++) {
    if (VAR == 0) BODY("s_add_u32 %0, %0, 1");                                   // dependent SALU, 4 bytes
    if (VAR == 1) BODY16("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1");  // independent SALU
    if (VAR == 2) BODY("s_add_u32 %0, %0, 0x12345");                             // dependent SALU, 8 bytes (literal)
    if (VAR == 3) BODY("v_add_u32 %4, %4, %5");                                  // dependent VALU, 4 bytes
    if (VAR == 4) BODY("v_add3_u32 %4, %4, %5, %5");                             // dependent VALU, 8 bytes (VOP3)
    if (VAR == 5) BODY32("s_add_u32 %0, %0, 1\n\tv_add_u32 %4, %4, %5");         // two independent chains, alternating units
    if (VAR == 6) BODY32("v_mov_b32 %4, %0\n\tv_readfirstlane_b32 %0, %4");      // scalar -> vector -> scalar hand-over
    if (VAR == 7) BODY("s_nop 0");
    if (VAR == 8) BODY32("v_add_u32 %4, %4, %5\n\tv_add_u32 %6, %6, %5");        // independent VALU
    if (VAR == 9) BODY16("s_and_b32 %1, %0, 0x1f8\n\ts_or_b32 %2, %1, %3\n\ts_lshr_b32 %1, %2, 3\n\ts_add_u32 %0, %0, %1");  // dependent SALU mix with a literal
    if (VAR == 10) BODY32("s_add_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\t");       // SALU + compare (SCC)
    if (VAR == 11) BODY32("s_add_u32 %0, %0, 1\n\tv_add_u32 %4, %4, %0");        // SALU result consumed by VALU
    if (VAR == 12) BODY32("ds_read_b32 %5, %6\n\ts_waitcnt lgkmcnt(0)");         // LDS round trip, same address
    if (VAR == 13) BODY16("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1");  // not-taken branch
    if (VAR == 14) BODY16("s_cmp_lg_u32 %1, 0\n\ts_cbranch_scc1 1\n\ts_add_u32 %1, %1, 0\n\ts_add_u32 %0, %0, 1");  // taken branch over one instruction
    // 16 instructions per repetition, 4 repetitions = 64
    if (VAR == 15) MEMBODY(TWELVE "s_mov_b32 m0, %8\n\ts_mov_b64 exec, 1\n\tglobal_load_lds_dword %9, %7\n\ts_mov_b64 exec, -1");  // the refill as it is
    if (VAR == 16) MEMBODY(TWELVE "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dword %9, %7\n\ts_nop 0");                    // ... without the EXEC writes (64 lanes load)
    if (VAR == 17) MEMBODY(TWELVE "ds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], %9 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");  // mark, entry read, take
    if (VAR == 18) MEMBODY("ds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], %9 offset1:1\n\t" TWELVE "s_waitcnt lgkmcnt(1)\n\ts_mov_b32 m0, %8\n\ts_mov_b64 exec, 1\n\tglobal_load_lds_dword %9, %7\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");  // 21 per repetition: the step's memory skeleton
    // 20/21: the skeleton with the refill's source moving through a 64 MiB table (4 KiB + 64 B steps: never the line just read)
    if (VAR == 20) MEMBODY("ds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], %9 offset1:1\n\t" TWELVE "v_add_u32 %6, 0x1040, %6\n\tv_and_b32 %6, 0x3ffffff, %6\n\ts_waitcnt lgkmcnt(1)\n\ts_mov_b32 m0, %8\n\ts_mov_b64 exec, 1\n\tglobal_load_lds_dword %6, %7\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 21) MEMBODY("ds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], %9 offset1:1\n\t" TWELVE "v_add_u32 %6, 0x1040, %6\n\tv_and_b32 %6, 0x3ffffff, %6\n\ts_waitcnt lgkmcnt(1)\n\ts_nop 0\n\ts_mov_b64 exec, 1\n\tglobal_load_dword v12, %6, %7\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 22) MEMBODY(TWELVE "v_add_u32 %6, 0x1040, %6\n\tv_and_b32 %6, 0x3ffffff, %6\n\ts_mov_b32 m0, %8\n\ts_mov_b64 exec, 1\n\tglobal_load_lds_dword %6, %7\n\ts_mov_b64 exec, -1");  // 18 per rep: far refills, no LDS operation of the wave's own
    // 23: the skeleton with the entry read's ADDRESS taken from the previous entry (readfirstlane -> s_and -> s_or -> v_mov -> ds_read): 24 per rep
    if (VAR == 23) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "s_waitcnt lgkmcnt(1)\n\ts_mov_b32 m0, %8\n\ts_mov_b64 exec, 1\n\tglobal_load_lds_dword %9, %7\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    // 24: ... and the twelve in between dependent on the entry too (s_add chains start from it)
    if (VAR == 24) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\ts_add_u32 %0, %1, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\tv_add_u32 %6, %0, %6\n\tv_and_b32 %6, 0xffc, %6\n\ts_waitcnt lgkmcnt(1)\n\ts_mov_b32 m0, %8\n\ts_mov_b64 exec, 1\n\tglobal_load_lds_dword %6, %7\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    // 25-28: the walker of the two-wave form: chain + a 16-byte packet written to the ring (all lanes, same address), no refill
    if (VAR == 25) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "v_mov_b32 v13, 0\n\tds_write_b128 v13, v[16:19] offset:1024\n\t" "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 26) { asm volatile("s_mov_b64 exec, 1"); MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "v_mov_b32 v13, 0\n\tds_write_b128 v13, v[16:19] offset:1024\n\t" "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10"); asm volatile("s_mov_b64 exec, -1"); }
    if (VAR == 27) { asm volatile("s_mov_b64 exec, 1"); MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10"); asm volatile("s_mov_b64 exec, -1"); }
    if (VAR == 28) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 29) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "v_mov_b32 v13, 0\n\tds_write_b32 v13, v16 offset:1024\n\t" "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 30) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "v_mov_b32 v13, 0\n\tds_write_b64 v13, v[16:17] offset:1024\n\t" "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 31) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "v_mov_b32 v13, 0\n\tds_write2_b32 v13, v16, v17 offset0:64 offset1:65\n\t" "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 32) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "v_mov_b32 v13, 0\n\tds_write_b96 v13, v[16:18] offset:1024\n\t" "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 33) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "v_mov_b32 v13, 0\n\tds_write2_b64 v13, v[16:17], v[18:19] offset0:128 offset1:129\n\t" "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 34) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\tv_mov_b32 v13, 0\n\tds_write_b128 v13, v[16:19] offset:1024\n\t" TWELVE "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 35) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, 0\n\tds_write_b128 v13, v[16:19] offset:1024\n\tds_write_b32 %9, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 36) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read_b32 v10, v12\n\t" TWELVE "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 37) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read_b64 v[10:11], v12\n\t" TWELVE "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 38) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %9, %5\n\tds_read_b32 v10, v12\n\tds_read_b32 v11, v12 offset:4\n\t" TWELVE "s_waitcnt lgkmcnt(1)\n\tv_readfirstlane_b32 %1, v10");
    if (VAR == 39) MEMBODY("s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t" TWELVE "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, v10");
    // 40-42 (round 4): the chain WITHOUT an LDS round trip on it -- the 64 candidate entries of the next context's symbol-free
    // half sit one per lane (read a step ahead), the symbol just decoded selects among them: s_bfe (symbol) -> v_readlane (next
    // entry).  The candidates' read (address from the entry of the step before) and everything else run in its shadow.
    if (VAR == 40) MEMBODY("s_bfe_u32 %2, %1, 0x60003\n\tv_readlane_b32 %1, v10, %2\n\ts_and_b32 %3, %1, 0x1f8\n\tv_or_b32 v12, %3, %9\n\tds_read_b32 v10, v12\n\t" TWELVE "s_waitcnt lgkmcnt(0)");   // 18 per rep
    // 41: ... plus the refill block (mark of the slot left + LDS-DMA): what the real step adds to it
    if (VAR == 41) MEMBODY("s_bfe_u32 %2, %1, 0x60003\n\tv_readlane_b32 %1, v10, %2\n\ts_and_b32 %3, %1, 0x1f8\n\tv_or_b32 v12, %3, %9\n\tds_write_b32 %9, %5\n\tds_read_b32 v10, v12\n\t" TWELVE "s_waitcnt lgkmcnt(1)\n\ts_mov_b32 m0, %8\n\ts_mov_b64 exec, 1\n\tglobal_load_lds_dword %9, %7\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)");  // 24 per rep
    // 42: the current step's skeleton with the same count of filler for comparison: 23 + readfirstlane = VAR 23
    if (VAR == 19) MEMBODY(TWELVE "s_mov_b32 m0, %8\n\ts_mov_b64 exec, 1\n\tglobal_load_dword v12, %9, %7\n\ts_mov_b64 exec, -1");  // a plain load instead of the LDS-DMA
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[blockIdx.x] = a + b + c + d + x + y + z;
    cyc[2 * blockIdx.x] = t1 - t0;
    cyc[2 * blockIdx.x + 1] = r1 - r0;
  }
}


// Two waves of one workgroup on the LDS: wave 0 runs the walker's skeleton (mark, entry read with the address from
// the entry, 12 scalar instructions, 16-byte packet write: 21 instructions a repetition), wave 1 a feeder's (4-byte
// write, 16-byte read, 14 vector instructions, wait).  MODE bit 0: wave 1 runs too; bit 1: both with lane 0 alone.
template <int MODE>
__global__ void __launch_bounds__(128) k2w(unsigned iters, unsigned *out, unsigned long long *cyc) {
  __shared__ unsigned lds[2048];
  for (unsigned i = threadIdx.x; i < 2048; i += 128) lds[i] = 0;
  __syncthreads();
  unsigned a = 1, b = 2, c = 3, d = 4, x = threadIdx.x, y = 7, z = 0, zero = 0;
  asm volatile("" : "+v"(zero));
  const bool w0 = __builtin_amdgcn_readfirstlane(threadIdx.x) < 64;
  if (!w0 && !(MODE & 1)) return;
  if (MODE & 2) asm volatile("s_mov_b64 exec, 1");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (w0) {
    for (unsigned i = 0; i < iters; i++)
      asm volatile(".rept 4\n\t"
                   "s_and_b32 %2, %1, 0x1f8\n\ts_or_b32 %2, %2, 0\n\tv_mov_b32 v12, %2\n\tds_write_b32 %7, %5\n\tds_read2_b32 v[10:11], v12 offset1:1\n\t"
                   "v_mov_b32 v13, 0\n\tds_write_b128 v13, v[16:19] offset:1024\n\t"
                   "s_add_u32 %0, %0, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %3, %3, 1\n\t"
                   "s_add_u32 %0, %0, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %3, %3, 1\n\t"
                   "s_waitcnt lgkmcnt(1)\n\tv_readfirstlane_b32 %1, v10\n\t.endr"
                   : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+v"(x), "+v"(y), "+v"(z) : "v"(zero) : "scc", "vcc", "memory", "v10", "v11", "v12", "v13", "v16", "v17", "v18", "v19");
  } else {
    for (unsigned i = 0; i < iters; i++)
      asm volatile(".rept 4\n\t"
                   "v_mov_b32 v13, 0\n\tds_write_b32 v13, %5 offset:2048\n\tds_read_b128 v[16:19], v13 offset:1024\n\t"
                   "v_add_u32 %4, %4, %5\n\tv_add_u32 %6, %6, %5\n\tv_add_u32 %4, %4, %5\n\tv_add_u32 %6, %6, %5\n\tv_add_u32 %4, %4, %5\n\tv_add_u32 %6, %6, %5\n\tv_add_u32 %4, %4, %5\n\t"
                   "v_add_u32 %6, %6, %5\n\tv_add_u32 %4, %4, %5\n\tv_add_u32 %6, %6, %5\n\tv_add_u32 %4, %4, %5\n\tv_add_u32 %6, %6, %5\n\tv_add_u32 %4, %4, %5\n\tv_add_u32 %6, %6, %5\n\t"
                   "s_waitcnt lgkmcnt(0)\n\tv_add_u32 %4, %4, v16\n\t.endr"
                   : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+v"(x), "+v"(y), "+v"(z) : "v"(zero) : "scc", "vcc", "memory", "v10", "v11", "v12", "v13", "v16", "v17", "v18", "v19");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (MODE & 2) asm volatile("s_mov_b64 exec, -1");
  if ((threadIdx.x & 63) == 0) {
    out[blockIdx.x * 2 + (w0 ? 0 : 1)] = a + b + c + d + x + y + z;
    cyc[blockIdx.x * 2 + (w0 ? 0 : 1)] = t1 - t0;
  }
}
template <int MODE> void run2(const char *name) {
  const unsigned iters = 20000;
  unsigned *out; unsigned long long *cyc;
  CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&cyc, 4096 * 16)); CK(hipMemset(cyc, 0, 4096 * 16));
  k2w<MODE><<<1, 128>>>(100, out, cyc);
  k2w<MODE><<<1, 128>>>(iters, out, cyc);
  CK(hipDeviceSynchronize());
  unsigned long long h[2]; CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
  printf("%-70s wave 0: %6.1f clocks a repetition, wave 1: %6.1f\n", name, (double)h[0] / (4.0 * iters), (double)h[1] / (4.0 * iters));
  CK(hipFree(out)); CK(hipFree(cyc));
}

template <int VAR> void run(const char *name, unsigned grid) {
  const unsigned iters = 20000;
  unsigned *out; unsigned long long *cyc;
  CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&cyc, 4096 * 16));
  unsigned *tab; CK(hipMalloc(&tab, 256u << 20)); CK(hipMemset(tab, 0, 256u << 20));
  k<VAR><<<grid, 64>>>(100, out, cyc, tab);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  k<VAR><<<grid, 64>>>(iters, out, cyc, tab);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2]; CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
  const double n = 64.0 * iters;
  printf("%-58s grid %4u: %6.2f ns/instr (wall)  %6.2f memtime ticks/instr  %6.2f realtime(100MHz) ticks/instr\n", name, grid, ms * 1e6 / n,
         (double)h[0] / n, (double)h[1] / n);
  CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(tab));
}

int main() {
  for (unsigned grid : {1u, 8u}) {
    run<0>("0 dependent s_add (4 B)", grid);
    run<1>("1 independent s_add x4 (4 B)", grid);
    run<2>("2 dependent s_add literal (8 B)", grid);
    run<3>("3 dependent v_add (4 B)", grid);
    run<4>("4 dependent v_add3 (8 B)", grid);
    run<5>("5 s_add | v_add alternating, independent", grid);
    run<6>("6 v_mov <- s ; v_readfirstlane (hand-over chain)", grid);
    run<7>("7 s_nop 0", grid);
    run<8>("8 independent v_add x2", grid);
    run<9>("9 dependent SALU mix with literal", grid);
    run<10>("10 s_add ; s_cmp", grid);
    run<11>("11 s_add -> v_add consumes it", grid);
    run<12>("12 ds_read ; wait (per pair: /2)", grid);
    run<13>("13 not-taken branch in 4 instr", grid);
    run<14>("14 taken branch over 1 instr (4 instr issued of 4... 3 executed)", grid);
    run<15>("15 12 s_add + refill (m0, exec, LDS-DMA, exec)", grid);
    run<16>("16 12 s_add + LDS-DMA without EXEC writes", grid);
    run<17>("17 12 s_add + mark, read, wait, readfirstlane", grid);
    run<18>("18 skeleton of the step: 21 instr per 16 counted (x 1.31)", grid);
    run<19>("19 12 s_add + plain global load under EXEC=1", grid);
    run<20>("20 skeleton, refills from far apart (23 instr per 16 counted)", grid);
    run<21>("21 skeleton, plain loads from far apart (23 per 16)", grid);
    run<23>("23 skeleton with the read address from the entry (24 per 16)", grid);
    run<24>("24 ... and the scalar work and refill address too (24 per 16)", grid);
    run<28>("28 chain + 12 s_add, no refill (19 per 16)", grid);
    run<27>("27 ... with lane 0 alone (EXEC = 1)", grid);
    run<25>("25 chain + 12 s_add + 16-byte packet write (21 per 16)", grid);
    run<26>("26 ... with lane 0 alone (EXEC = 1)", grid);
    run<29>("29 chain + second ds_write_b32 (21 per 16)", grid);
    run<30>("30 chain + ds_write_b64 (21 per 16)", grid);
    run<31>("31 chain + ds_write2_b32 (21 per 16)", grid);
    run<32>("32 chain + ds_write_b96 (21 per 16)", grid);
    run<33>("33 chain + ds_write2_b64 (21 per 16)", grid);
    run<34>("34 chain + b128 right behind the entry read (21 per 16)", grid);
    run<35>("35 chain + b128 in front of mark and read (21 per 16)", grid);
    run<36>("36 chain (28) with ds_read_b32", grid);
    run<37>("37 chain (28) with ds_read_b64", grid);
    run<38>("38 chain (28) with two ds_read_b32, the wait for the first only (20 per 16)", grid);
    run<39>("39 chain (28) without the mark (18 per 16)", grid);
    run<22>("22 12 s_add + far refills, no ds ops (18 per 16)", grid);
    run<40>("40 readlane-select chain + candidates' read + 12 s_add (18 per 16)", grid);
    run<41>("41 ... + mark and refill (24 per 16)", grid);
  }
  run2<0>("two waves: walker skeleton alone");
  run2<1>("two waves: walker and feeder skeletons");
  run2<2>("two waves: walker alone, lane 0 only");
  run2<3>("two waves: walker and feeder, lane 0 only");
  return 0;
}
