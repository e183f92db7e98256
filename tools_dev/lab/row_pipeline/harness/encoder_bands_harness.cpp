// Host harness for csrc/encoder_bands.hip (tests/test_encoder_bands_host.py): the launcher object is linked against THIS file
// instead of the HIP runtime and the library's kernels, so its launch sequence — which entry point, on which stream, with
// which pointers, behind which events — can be checked without a GPU.  Every call is appended to a text trace.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

typedef void* hipEvent_t;
typedef void* hipStream_t;
typedef int hipError_t;

static std::string g_trace;
static std::string g_err;
static int g_events = 0;
static int g_fail_at = -1, g_calls = 0;      // make the g_fail_at-th kernel entry fail (error-path test)

static void add(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_trace += buf;
}
static int kernel_rc() { return (g_fail_at >= 0 && g_calls++ == g_fail_at) ? -2 : 0; }

namespace occ {
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
}  // namespace occ

extern "C" {
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
  *e = reinterpret_cast<hipEvent_t>(static_cast<intptr_t>(0xE0000 + ++g_events));
  return 0;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { add("record %p %p\n", e, s); return 0; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { add("wait %p %p\n", s, e); return 0; }
hipError_t hipStreamSynchronize(hipStream_t s) { add("sync %p\n", s); return 0; }
const char* hipGetErrorString(hipError_t) { return "harness"; }

int occ_tsa_fused_forward_f32(const float* value, int64_t value_bt_stride, const float* offs, int64_t offs_stride,
                              const float* logits, int64_t logits_stride, const float* ref_2d, const int32_t* order,
                              float* out, int B, int Nq, int bev_h, int bev_w, int M, int D, int P, void* stream) {
  add("T %p value=%p vstride=%ld offs=%p os=%ld logits=%p ls=%ld ref=%p order=%p out=%p B=%d n=%d bev=%dx%d M=%d D=%d P=%d\n",
      stream, (const void*)value, (long)value_bt_stride, (const void*)offs, (long)offs_stride, (const void*)logits,
      (long)logits_stride, (const void*)ref_2d, (const void*)order, (void*)out, B, Nq, bev_h, bev_w, M, D, P);
  return kernel_rc();
}
int occ_linear_ln_chain_bf16x3_f32(const float* a, int64_t lda, const float* residual, int64_t ldres, const void* w_chain,
                                   const float* bias_chain, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                   float* y, int64_t ldy, float* z, int64_t ldz, int n2, int act2, int M, void* stream) {
  add("A %p a=%p lda=%ld res=%p ldres=%ld w=%p bias=%p g=%p b=%p eps=%g y=%p ldy=%ld z=%p ldz=%ld n2=%d act=%d n=%d\n", stream,
      (const void*)a, (long)lda, (const void*)residual, (long)ldres, w_chain, (const void*)bias_chain, (const void*)ln_gamma,
      (const void*)ln_beta, (double)ln_eps, (void*)y, (long)ldy, (void*)z, (long)ldz, n2, act2, M);
  return kernel_rc();
}
static int sca(const char* tag, const void* value, const float* offs, int64_t offs_stride, const float* logits,
               int64_t logits_stride, const float* ref_cam, const uint32_t* vis_bits, const int32_t* order, float* slots,
               uint64_t* stats, int B, int NC, int S, int M, int D, int L, int P, int Z, int Nq, void* stream) {
  add("%s %p value=%p offs=%p os=%ld logits=%p ls=%ld ref=%p vis=%p order=%p slots=%p stats=%p B=%d NC=%d S=%d M=%d D=%d L=%d "
      "P=%d Z=%d n=%d\n", tag, stream, value, (const void*)offs, (long)offs_stride, (const void*)logits, (long)logits_stride,
      (const void*)ref_cam, (const void*)vis_bits, (const void*)order, (void*)slots, (void*)stats, B, NC, S, M, D, L, P, Z, Nq);
  return kernel_rc();
}
int occ_sca_fused_forward_f16v(const void* value, const int64_t*, const int64_t*, const float* offs, int64_t offs_stride,
                               const float* logits, int64_t logits_stride, const float* ref_cam, const uint32_t* vis_bits,
                               const int32_t* order, float* slots, uint64_t* stats, int B, int NC, int S, int M, int D, int L,
                               int P, int Z, int Nq, const float* /*value_scale*/, void* stream) {
  return sca("S16", value, offs, offs_stride, logits, logits_stride, ref_cam, vis_bits, order, slots, stats, B, NC, S, M, D, L,
             P, Z, Nq, stream);
}
int occ_sca_fused_forward_f32(const float* value, const int64_t*, const int64_t*, const float* offs, int64_t offs_stride,
                              const float* logits, int64_t logits_stride, const float* ref_cam, const uint32_t* vis_bits,
                              const int32_t* order, float* slots, uint64_t* stats, int B, int NC, int S, int M, int D, int L,
                              int P, int Z, int Nq, void* stream) {
  return sca("S32", value, offs, offs_stride, logits, logits_stride, ref_cam, vis_bits, order, slots, stats, B, NC, S, M, D, L,
             P, Z, Nq, stream);
}
int occ_encoder_ffn_chain_bf16x3_f32(const float* a, int64_t lda, const float* residual, int64_t ldres, const void* w_chain,
                                     const float* bias_chain, const float* ln1_gamma, const float* ln1_beta, float ln1_eps,
                                     const float* ln2_gamma, const float* ln2_beta, float ln2_eps, float* y, int64_t ldy,
                                     const float* q_term, int64_t ldq_term, float* zq, int64_t ldzq, int nq, float* zv,
                                     int64_t ldzv, int M, void* stream) {
  add("B %p a=%p res=%p w=%p bias=%p g1=%p b1=%p eps1=%g g2=%p b2=%p eps2=%g y=%p ldy=%ld qterm=%p ldq=%ld zq=%p ldzq=%ld nq=%d "
      "zv=%p ldzv=%ld n=%d lda=%ld ldres=%ld\n", stream, (const void*)a, (const void*)residual, w_chain, (const void*)bias_chain,
      (const void*)ln1_gamma, (const void*)ln1_beta, (double)ln1_eps, (const void*)ln2_gamma, (const void*)ln2_beta,
      (double)ln2_eps, (void*)y, (long)ldy, (const void*)q_term, (long)ldq_term, (void*)zq, (long)ldzq, nq, (void*)zv, (long)ldzv,
      M, (long)lda, (long)ldres);
  return kernel_rc();
}

const char* harness_trace(void) { return g_trace.c_str(); }
const char* harness_error(void) { return g_err.c_str(); }
void harness_reset(int fail_at) {
  g_trace.clear();
  g_err.clear();
  g_fail_at = fail_at;
  g_calls = 0;
}
}
