// Host side of gemm_tile_p8_kernel (vg_gemm_p8.hip, compiled four times: bf16 / fp32 output x whole-tile / stream-K): eligibility, the stream-K
// launch plan and its per-stream workspace, and the dispatch to the instantiation.
#include "vg_gemm_common.h"
#include <mutex>
#include <unordered_map>

int vg_p8_launch_bf16(const GemmArgs& q, int wgs, hipStream_t st);
int vg_p8_launch_bf16_sk(const GemmArgs& q, int wgs, hipStream_t st);
int vg_p8_launch_f32(const GemmArgs& q, int wgs, hipStream_t st);
int vg_p8_launch_f32_sk(const GemmArgs& q, int wgs, hipStream_t st);

// Launcher (vg_gemm.hip's route_w128 decides; this only checks what the 32-bit source offsets need)
bool vg_gemm_p8_eligible(const GemmArgs& p, int batch) {
  const int64_t arows = p.M, wrows = p.a_op == 1 ? 2 * (int64_t)p.N : p.N;
  return p.K % 64 == 0 && p.K >= 128 && arows * p.lda * 2 < (int64_t)1 << 32 && wrows * p.ldw * 2 < (int64_t)1 << 32 && !p.sa && !p.wmode && p.vec_out;
}

// Stream-K workspace: one per stream that has launched a stream-K GEMM (kernels of ONE stream run back to back, kernels of different streams —
// Hiera beside the LLM prefill, the two vision towers — may overlap and must not share partial slots, flags or the ticket).  Allocated on first use
// (never during a stream capture: a capturing stream takes the whole-tile form), kept for the life of the process.
namespace {
int env_knob(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
struct SkWs { float* part; int* ctl; };
SkWs* sk_workspace(hipStream_t st, int ncu) {
  static std::mutex mu;
  static std::unordered_map<hipStream_t, SkWs> table;
  std::lock_guard<std::mutex> lock(mu);
  auto it = table.find(st);
  if (it != table.end()) return &it->second;
  if (table.size() >= 16) return nullptr;       // (a bound on 64 MB slots; the path uses four streams)
  SkWs w{nullptr, nullptr};
  const size_t ctl_bytes = (size_t)(SK_FLAG0 + ncu * 8 * SK_FLAG_STRIDE) * sizeof(int);
  if (hipMalloc((void**)&w.part, (size_t)ncu * 256 * 256 * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMalloc((void**)&w.ctl, ctl_bytes) != hipSuccess || hipMemset(w.ctl, 0, ctl_bytes) != hipSuccess) {
    (void)hipFree(w.part);
    return nullptr;
  }
  (void)hipDeviceSynchronize();
  return &table.emplace(st, w).first->second;
}
}  // namespace

// Plan.  Whole rounds of `ncu` tiles leave the last round partly empty; cutting the tail by K steps instead saves
// (rounds_up - tot / ncu) * nk K steps (about 1.65 us each) per workgroup and costs one publish (256 KB of write-through stores) and one fix-up
// (a fabric round trip + 256 KB of reads) — about 9 us: stream-K is taken from VG_GEMM_SK_MINSAVE (default 10) saved K steps on.
//   tot >= ncu: the last WHOLE round and the remainder are cut together ("two-tile": spans of nk ... 2 nk steps, a tile is cut at most once);
//   tot <  ncu: everything is cut, over min(ncu, 2 tot) workgroups: spans >= nk / 2, a tile meets at most three spans (two successors — what
//               the fix-up's LDS staging holds).
// the plan of a shape: 0 = whole tiles only; else the number of workgroups of the stream-K launch (*dp_rounds whole rounds in front of the spans)
int vg_gemm_p8_sk_plan(int64_t M, int64_t N, int64_t K, int a_op, int batch, int* dp_rounds) {
  static const int sk_mode = env_knob("VG_GEMM_SK", 1);
  static const int sk_minsave = env_knob("VG_GEMM_SK_MINSAVE", 10);
  static const int sk_minsteps = env_knob("VG_GEMM_SK_MINK", 8);
  static const int ncu = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  const int64_t nk = K / 64;
  const int64_t mt = (M + 255) / 256, nt = a_op == 1 ? (N + 127) / 128 : (N + 255) / 256;
  const int64_t tot = mt * nt * batch;
  if (!sk_mode || K % 64 != 0 || nk % 2 != 0 || nk < sk_minsteps || tot % ncu == 0) return 0;
  const int64_t rounds = tot / ncu;
  const int sk_wgs = rounds > 0 ? ncu : (int)(2 * tot < ncu ? 2 * tot : ncu);
  const double saved = (double)(rounds + 1) * nk - (rounds > 0 ? (double)tot * nk / ncu : (double)tot * nk / sk_wgs);
  if (saved < sk_minsave) return 0;
  if (dp_rounds) *dp_rounds = rounds > 0 ? (int)rounds - 1 : 0;
  return sk_wgs;
}

int vg_gemm_p8_launch(const GemmArgs& q0, int out_is_bf16, int wgs, hipStream_t st) {
  GemmArgs q = q0;
  q.sk = 0;
  int dp = 0;
  if (const int sk_wgs = vg_gemm_p8_sk_plan(q.M, q.N, q.K, q.a_op, q.nbatch, &dp)) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
      static const int ncu = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
      if (SkWs* w = sk_workspace(st, ncu)) {
        q.sk = 1;
        q.sk_dp = dp;
        q.sk_part = w->part;
        q.sk_ctl = w->ctl;
        wgs = sk_wgs;
      }
    }
  }
  if (q.sk) return out_is_bf16 ? vg_p8_launch_bf16_sk(q, wgs, st) : vg_p8_launch_f32_sk(q, wgs, st);
  return out_is_bf16 ? vg_p8_launch_bf16(q, wgs, st) : vg_p8_launch_f32(q, wgs, st);
}
