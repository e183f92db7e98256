// Native launch sequence of the BEV encoder's chain path: every layer's TSA gather -> chain program A -> SCA gather -> chain
// program B, for K row bands of the BEV queries on K HIP streams, issued from ONE C-ABI call.
//
// Why this exists: between two TSA gathers a BEVFormerLayer is row-local (reference: encoder.py:377-404 — the TSA output
// projection, the SCA of a query and the FFN touch that query's row only; spatial_cross_attention.py:136-173 samples the
// CAMERA planes, temporal_self_attention.py:240-262 is the one place that reads the whole BEV), so two bands of queries can
// be in different kernels at the same time: band 2's TSA gather / program A under band 1's SCA gather, band 1's program B
// under band 2's SCA gather.  Driven from Python (plugin/encoder.py, OCC_ENCODER_ROW_PIPELINE) the banded sequence is
// host-bound — 2.44 ms of launch-side time per step against 1.0 ms for the unbanded one, profiles/r04_rowpipe_host.log —
// so its wall time says nothing about the overlap.  Here the whole sequence costs one ctypes call: 4 kernel launches, one
// event record and K - 1 event waits per band and layer, all from C++.  With K = 1 it is the unbanded chain path on the
// caller's stream (the same launches plugin/encoder.py issues one by one).
//
// STATUS: written at the end of round 4 after the GPU budget was spent — compiled, argument checks tested on the host,
// NEVER RUN on an MI355X.  Opt-in (OCC_ENCODER_ROW_PIPELINE=K with OCC_ROW_PIPELINE_NATIVE=1); tests/test_gpu_row_pipeline.py
// is its parity test.
//
// flags (experiments for the round that first runs this): OCC_EB_STAGGER — band i's very first launch waits for band
// i - 1's first launch, so the bands run one stage apart and DIFFERENT kernels share the chip (no per-stage events: the
// Python pipeline's per-stage serialisation produced wrong rows, plugin/encoder.py); OCC_EB_BAND_MAJOR — submission order.
//
// The kernels are the library's own entry points (sca_fused.hip, tsa_fused.hip, linear_chain_x3.hip), called with band
// pointers: at B = 1 the gathers index offsets / logits / reference points / visibility / output by query only and take a
// band-local processing order, the chain kernels take row pointers (see OccBand in include/occnet_amd.h).
#include <vector>
#include "common.h"

namespace occ {
namespace {

// events of one calling thread, reused from call to call (a wait captures the event's state when it is issued, so a later
// re-record cannot disturb it)
struct EventPool {
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  hipEvent_t get() {
    if (used == ev.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      ev.push_back(e);
    }
    return ev[used++];
  }
};
thread_local EventPool g_events;

}  // namespace
}  // namespace occ

extern "C" int occ_encoder_bands_forward_f32(const float* q0, const float* zq0, int64_t ldzq0, const float* zv0,
                                             const OccBandLayer* layers, int n_layers, const OccBand* bands,
                                             int n_bands, const int64_t* spatial_shapes,
                                             const int64_t* level_start_index, const uint32_t* vis_bits, int Nq,
                                             int bev_h, int bev_w, int NC, int S, int L, int P, int Z, int tsa_P,
                                             int planes_f16, int flags, void* main_stream) {
  using namespace occ;
  constexpr int M = 8, D = 32, C = 256;
  OCC_CHECK_ARG(q0 && zq0 && zv0 && layers && bands && spatial_shapes && level_start_index && vis_bits,
                "encoder_bands_forward: null pointer argument");
  OCC_CHECK_ARG(n_layers > 0 && n_bands > 0 && n_bands <= 16 && Nq > 0 && bev_h > 0 && bev_w > 0 &&
                    (long)bev_h * bev_w == Nq && NC > 0 && S > 0 && L > 0 && P > 0 && Z > 0 && tsa_P > 0,
                "encoder_bands_forward: bad dimension (layers=%d bands=%d Nq=%d bev=%dx%d NC=%d S=%d L=%d P=%d Z=%d)",
                n_layers, n_bands, Nq, bev_h, bev_w, NC, S, L, P, Z);
  const int tsa_noff = M * 2 * tsa_P * 2, tsa_natt = M * 2 * tsa_P;        // TSA query Linears: offsets | weights
  const int sca_noff = M * L * P * 2, n_lin = M * L * P * 3;                 // SCA query Linears: offsets | weights
  OCC_CHECK_ARG(ldzq0 >= tsa_noff + tsa_natt, "encoder_bands_forward: zq0 rows shorter than offsets + weights");
  {
    long next = 0;                                                            // the bands tile [0, Nq) in order
    for (int i = 0; i < n_bands; ++i) {
      const OccBand& b = bands[i];
      OCC_CHECK_ARG(b.m0 == next && b.n > 0, "encoder_bands_forward: band %d does not continue the previous one", i);
      OCC_CHECK_ARG(b.order && b.ref_2d && b.ref_cam && b.attn && b.x1 && b.lin && b.slots,
                    "encoder_bands_forward: band %d has a null pointer", i);
      next += b.n;
    }
    OCC_CHECK_ARG(next == Nq, "encoder_bands_forward: the bands cover %ld of %d queries", next, Nq);
  }
  for (int l = 0; l < n_layers; ++l) {
    const OccBandLayer& y = layers[l];
    OCC_CHECK_ARG(y.wA && y.biasA && y.ln0_g && y.ln0_b && y.plane && y.wB && y.biasB && y.ln1_g && y.ln1_b && y.ln2_g &&
                      y.ln2_b && y.out,
                  "encoder_bands_forward: layer %d has a null pointer", l);
    const bool tail = y.zq != nullptr || y.zv != nullptr;
    OCC_CHECK_ARG(tail == (l + 1 < n_layers), "encoder_bands_forward: every layer but the last needs the tail outputs");
    OCC_CHECK_ARG(!tail || (y.zq && y.zv && y.nq_tail >= tsa_noff + tsa_natt && y.nq_tail <= 256 &&
                            (!y.q_term || y.ldq_term >= y.nq_tail)),
                  "encoder_bands_forward: layer %d: bad tail", l);
  }
  hipStream_t main = reinterpret_cast<hipStream_t>(main_stream);
  EventPool& pool = g_events;
  pool.used = 0;
  int rc = OCC_OK;
#define OCC_EB_HIP(call)                                                                           \
  if (rc == OCC_OK) {                                                                              \
    const hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                       \
      set_error("encoder_bands_forward: %s failed: %s", #call, hipGetErrorString(e__));            \
      rc = OCC_E_LAUNCH;                                                                           \
    }                                                                                              \
  }
  // ---- fork: every band stream starts behind the caller's stream ------------------------------------------------------
  hipEvent_t fork = pool.get();
  if (!fork) {
    set_error("encoder_bands_forward: hipEventCreate failed");
    return OCC_E_LAUNCH;
  }
  OCC_EB_HIP(hipEventRecord(fork, main));
  std::vector<hipStream_t> st(n_bands);
  for (int i = 0; i < n_bands; ++i) {
    st[i] = reinterpret_cast<hipStream_t>(bands[i].stream);
    if (st[i] != main) OCC_EB_HIP(hipStreamWaitEvent(st[i], fork, 0));
  }
  std::vector<hipEvent_t> prev_b(n_bands, nullptr), cur_b(n_bands, nullptr);
  const float* q_prev = q0;                 // the layer input rows (Nq, 256): residual of program A
  const float* zq = zq0;                    // the TSA query Linears' outputs of this layer (Nq, ldzq)
  long ldzq = ldzq0;
  const float* zv = zv0;                    // this layer's projected BEV (Nq, 256): the TSA gather's value map
  hipEvent_t first_done = nullptr;          // OCC_EB_STAGGER: the previous band's first launch
  for (int l = 0; l < n_layers && rc == OCC_OK; ++l) {
    const OccBandLayer& y = layers[l];
    const bool tail = l + 1 < n_layers;
    // submission order: stage-major (T of every band, then A, S, B: the bands' kernels of one kind reach the dispatcher
    // together) or, OCC_EB_BAND_MAJOR, band-major (all four stages of band 0, then band 1, ...)
    for (int step = 0; step < 4 * n_bands && rc == OCC_OK; ++step) {
      const int stage = (flags & OCC_EB_BAND_MAJOR) ? step % 4 : step / n_bands;
      const int i = (flags & OCC_EB_BAND_MAJOR) ? step / 4 : step % n_bands;
      const OccBand& b = bands[i];
      if (stage == 0) {            // T: TSA gather of the band against the WHOLE BEV of the layer before
        if (l > 0)
          for (int j = 0; j < n_bands; ++j)
            if (j != i && st[j] != st[i]) OCC_EB_HIP(hipStreamWaitEvent(st[i], prev_b[j], 0));
        if (l == 0 && i > 0 && (flags & OCC_EB_STAGGER) && first_done && st[i] != st[i - 1])
          OCC_EB_HIP(hipStreamWaitEvent(st[i], first_done, 0));      // one stage behind the band before, once
        if (rc != OCC_OK) break;
        const float* lin = zq + (long)b.m0 * ldzq;
        rc = occ_tsa_fused_forward_f32(zv, 0, lin, ldzq, lin + tsa_noff, ldzq, b.ref_2d, b.order, b.attn, 1, b.n, bev_h,
                                       bev_w, M, D, tsa_P, st[i]);
        if (rc == OCC_OK && l == 0 && (flags & OCC_EB_STAGGER) && i + 1 < n_bands) {
          first_done = pool.get();
          if (!first_done) {
            set_error("encoder_bands_forward: hipEventCreate failed");
            rc = OCC_E_LAUNCH;
            break;
          }
          OCC_EB_HIP(hipEventRecord(first_done, st[i]));
        }
      } else if (stage == 1) {     // A: output_proj + LN -> the SCA's query Linears
        rc = occ_linear_ln_chain_bf16x3_f32(b.attn, C, q_prev + (long)b.m0 * C, C, y.wA, y.biasA, y.ln0_g, y.ln0_b,
                                            y.ln0_eps, b.x1, C, b.lin, n_lin, n_lin, 0, b.n, st[i]);
      } else if (stage == 2) {     // S: SCA gather of the band's queries
        if (y.plane_ready) OCC_EB_HIP(hipStreamWaitEvent(st[i], reinterpret_cast<hipEvent_t>(y.plane_ready), 0));
        if (rc != OCC_OK) break;
        rc = planes_f16
                 ? occ_sca_fused_forward_f16v(y.plane, spatial_shapes, level_start_index, b.lin, n_lin, b.lin + sca_noff,
                                              n_lin, b.ref_cam, vis_bits + b.m0, b.order, b.slots, y.stats, 1, NC, S, M, D, L,
                                              P, Z, b.n, y.plane_scale, st[i])
                 : occ_sca_fused_forward_f32(reinterpret_cast<const float*>(y.plane), spatial_shapes, level_start_index,
                                             b.lin, n_lin, b.lin + sca_noff, n_lin, b.ref_cam, vis_bits + b.m0, b.order,
                                             b.slots, y.stats, 1, NC, S, M, D, L, P, Z, b.n, st[i]);
      } else {                     // B: output_proj + LN + FFN + LN (+ the next layer's TSA Linears)
        rc = occ_encoder_ffn_chain_bf16x3_f32(
            b.slots, C, b.x1, C, y.wB, y.biasB, y.ln1_g, y.ln1_b, y.ln1_eps, y.ln2_g, y.ln2_b, y.ln2_eps,
            y.out + (long)b.m0 * C, C, (tail && y.q_term) ? y.q_term + (long)b.m0 * y.ldq_term : nullptr, y.ldq_term,
            tail ? y.zq + (long)b.m0 * y.nq_tail : nullptr, y.nq_tail, tail ? y.nq_tail : 0,
            tail ? y.zv + (long)b.m0 * C : nullptr, C, b.n, st[i]);
        if (rc == OCC_OK && tail && n_bands > 1) {
          cur_b[i] = pool.get();
          if (!cur_b[i]) {
            set_error("encoder_bands_forward: hipEventCreate failed");
            rc = OCC_E_LAUNCH;
            break;
          }
          OCC_EB_HIP(hipEventRecord(cur_b[i], st[i]));
        }
      }
    }
    prev_b.swap(cur_b);
    q_prev = y.out;
    if (tail) {
      zq = y.zq;
      ldzq = y.nq_tail;
      zv = y.zv;
    }
  }
  // ---- join (also after a failure: nothing enqueued on a band stream may outlive this call unordered) ----------------
  for (int i = 0; i < n_bands; ++i) {
    if (st[i] == main) continue;
    hipEvent_t e = pool.get();
    if (e && hipEventRecord(e, st[i]) == hipSuccess && hipStreamWaitEvent(main, e, 0) == hipSuccess) continue;
    (void)hipStreamSynchronize(st[i]);      // last resort: order by waiting
  }
#undef OCC_EB_HIP
  return rc;
}
