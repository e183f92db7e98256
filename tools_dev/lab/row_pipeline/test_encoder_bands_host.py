"""not-gpu: the launch sequence of csrc/encoder_bands.hip (occ_encoder_bands_forward_f32, the native launcher of the encoder's
row pipeline) WITHOUT a GPU.

The launcher's compiled object (occnet_amd/lib/obj/encoder_bands.o: host code only, it contains no kernel) is linked against
tests/harness/encoder_bands_harness.cpp instead of the HIP runtime and the library's kernel entry points; the harness writes
every call into a trace.  Checked here:
  * which entry point runs for which band of which layer, on the band's stream, with the band's rows of every buffer
    (pointer arithmetic in bytes);
  * the happens-before relation the events build: a band's TSA gather is ordered behind EVERY band's program B of the layer
    before (it reads the whole projected BEV), the caller's stream is ordered behind everything when the call returns, no
    wait refers to an event that was not recorded, every gather waits for the planes' event;
  * a failing launch still joins the streams and reports the failure.
The kernels themselves and the real runtime are exercised by tests/test_gpu_row_pipeline.py (opt-in)."""
import ctypes
import os
import subprocess

import pytest

from occnet_amd import build as occ_build, ext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    obj = os.environ.get("OCC_EB_TEST_OBJ") or os.path.join(occ_build.OBJDIR, "encoder_bands.o")   # (override: mutation checks)
    if not os.path.exists(obj):
        occ_build.build()
    so = tmp_path_factory.mktemp("eb") / "eb_harness.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", str(so),
                    os.path.join(ROOT, "tests", "harness", "encoder_bands_harness.cpp"), obj], check=True)
    lib = ctypes.CDLL(str(so))
    lib.harness_trace.restype = ctypes.c_char_p
    lib.harness_error.restype = ctypes.c_char_p
    return lib


NQ, BH, BW, NC, S, L, P, Z, TSA_P = 40000, 200, 200, 6, 30826, 4, 8, 4, 4
N_LIN, TSA_NOFF, NQ_TAIL = 8 * L * P * 3, 8 * 2 * TSA_P * 2, 192
MAIN = 0x111000


def _addr(kind, i=0, j=0):
    """distinct fake device addresses, far apart"""
    return (0x10 + kind) * 0x100000000 + i * 0x10000000 + j * 0x1000000


def _setup(n_layers, cuts, f16=True, same_stream=False):
    K = len(cuts) - 1
    bands = (ext._OccBand * K)()
    for i, b in enumerate(bands):
        b.m0, b.n = cuts[i], cuts[i + 1] - cuts[i]
        b.order, b.ref_2d, b.ref_cam = _addr(1, i), _addr(2, i), _addr(3, i)
        b.attn, b.x1, b.lin, b.slots = _addr(4, i), _addr(5, i), _addr(6, i), _addr(7, i)
        b.stream = MAIN if (K == 1 or same_stream) else 0x222000 + i * 0x1000
    layers = (ext._OccBandLayer * n_layers)()
    for l, y in enumerate(layers):
        y.wA, y.biasA, y.ln0_g, y.ln0_b, y.ln0_eps = _addr(8, l), _addr(9, l), _addr(10, l), _addr(11, l), 1e-5
        y.plane, y.plane_ready, y.stats = _addr(12, l), 0xABC000, None
        y.wB, y.biasB, y.ln1_g, y.ln1_b, y.ln1_eps = _addr(13, l), _addr(14, l), _addr(15, l), _addr(16, l), 2e-5
        y.ln2_g, y.ln2_b, y.ln2_eps = _addr(17, l), _addr(18, l), 3e-5
        y.out = _addr(19, l)
        if l + 1 < n_layers:
            y.q_term, y.ldq_term, y.nq_tail, y.zq, y.zv = _addr(20, l), NQ_TAIL, NQ_TAIL, _addr(21, l), _addr(22, l)
    return layers, bands


def _call(lib, layers, bands, f16=True, fail_at=-1, flags=0):
    lib.harness_reset(fail_at)
    q0, zq0, zv0 = _addr(23), _addr(24), _addr(25)
    rc = lib.occ_encoder_bands_forward_f32(
        ctypes.c_void_p(q0), ctypes.c_void_p(zq0), ctypes.c_int64(NQ_TAIL), ctypes.c_void_p(zv0), layers, len(layers), bands,
        len(bands), ctypes.c_void_p(_addr(26)), ctypes.c_void_p(_addr(27)), ctypes.c_void_p(_addr(28)), NQ, BH, BW, NC, S, L, P, Z,
        TSA_P, 1 if f16 else 0, flags, ctypes.c_void_p(MAIN))
    ops = []
    for line in lib.harness_trace().decode().splitlines():
        head, *rest = line.split()
        if head in ("record", "wait", "sync"):
            ops.append((head,) + tuple(int(v, 16) for v in rest))
        else:
            kv = dict(f.split("=") for f in rest[1:])
            ops.append((head, int(rest[0], 16) if rest[0] != "(nil)" else 0,
                        {k: (int(v, 16) if v.startswith("0x") else 0 if v == "(nil)" else v) for k, v in kv.items()}))
    return rc, ops, (q0, zq0, zv0)


def _happens_before(ops):
    """-> (kernels, after): kernels = [(index, tag, stream, args)], after[i] = set of kernel indices ordered before op i."""
    last_on = {}            # stream -> set of kernels ordered before the stream's next op
    snap = {}               # event -> that set at record time
    kernels, after = [], {}
    for idx, op in enumerate(ops):
        if op[0] == "record":
            snap[op[1]] = set(last_on.get(op[2], set()))
        elif op[0] == "wait":
            assert op[2] in snap or op[2] == 0xABC000, f"wait on an event that was never recorded: {op}"
            last_on.setdefault(op[1], set()).update(snap.get(op[2], set()))
        elif op[0] == "sync":
            pass
        else:
            tag, stream, args = op
            after[idx] = set(last_on.get(stream, set()))
            kernels.append((idx, tag, stream, args))
            last_on.setdefault(stream, set()).add(idx)
    return kernels, after, last_on


@pytest.mark.parametrize("cuts,n_layers,f16,flags", [
    ((0, 20800, 40000), 4, True, 0), ((0, 12800, 27200, 40000), 3, True, 0), ((0, 40000), 4, True, 0),
    ((0, 20800, 40000), 2, False, 0), ((0, 20800, 40000), 3, True, 1), ((0, 12800, 27200, 40000), 3, True, 2),
    ((0, 12800, 27200, 40000), 2, True, 3), ((0, 40000), 2, True, 3)])
def test_launch_sequence_pointers_and_ordering(harness, cuts, n_layers, f16, flags):
    layers, bands = _setup(n_layers, cuts, f16)
    rc, ops, (q0, zq0, zv0) = _call(harness, layers, bands, f16, flags=flags)
    assert rc == 0, harness.harness_error()
    K = len(bands)
    kernels, after, last_on = _happens_before(ops)
    assert len(kernels) == 4 * K * n_layers
    sca_tag = "S16" if f16 else "S32"
    by = {}
    pos = 0
    tags = ("T", "A", sca_tag, "B")
    for l in range(n_layers):                                   # submission order: stage-major, or band-major (flag 2)
        order = [(tag, i) for i in range(K) for tag in tags] if flags & 2 else [(tag, i) for tag in tags for i in range(K)]
        for tag, i in order:
            idx, t, stream, a = kernels[pos]
            pos += 1
            assert t == tag and stream == bands[i].stream, (l, tag, i, t, hex(stream))
            assert int(a["n"]) == bands[i].n
            by[(l, tag[0], i)] = (idx, a)
    for i in range(1, K):                                       # flag 1: one stage apart — band i's first launch is
        staggered = by[(0, "T", i - 1)][0] in after[by[(0, "T", i)][0]]      # ordered behind band i - 1's first launch
        assert staggered == bool(flags & 1)
    f4 = 4
    for l in range(n_layers):
        y = layers[l]
        tail = l + 1 < n_layers
        q_prev = q0 if l == 0 else layers[l - 1].out
        zq, ldzq, zv = (zq0, NQ_TAIL, zv0) if l == 0 else (layers[l - 1].zq, layers[l - 1].nq_tail, layers[l - 1].zv)
        for i, b in enumerate(bands):
            idx, a = by[(l, "T", i)]
            lin = zq + b.m0 * ldzq * f4
            assert (a["value"], a["vstride"], a["offs"], a["logits"]) == (zv, "0", lin, lin + TSA_NOFF * f4)
            assert (a["os"], a["ls"]) == (str(ldzq), str(ldzq)) and (a["ref"], a["order"], a["out"]) == (b.ref_2d, b.order, b.attn)
            assert (a["B"], a["bev"], a["M"], a["D"], a["P"]) == ("1", f"{BH}x{BW}", "8", "32", str(TSA_P))
            if l > 0:                                           # behind EVERY band's program B of the layer before
                assert {by[(l - 1, "B", j)][0] for j in range(K)} <= after[idx]
            idx, a = by[(l, "A", i)]
            assert (a["a"], a["res"], a["w"], a["bias"], a["g"], a["b"]) == (b.attn, q_prev + b.m0 * 256 * f4, y.wA, y.biasA,
                                                                          y.ln0_g, y.ln0_b)
            assert (a["y"], a["z"], a["ldz"], a["n2"], a["act"], a["lda"], a["ldres"], a["ldy"]) == (
                b.x1, b.lin, str(N_LIN), str(N_LIN), "0", "256", "256", "256")
            assert by[(l, "T", i)][0] in after[idx]
            idx, a = by[(l, "S", i)]
            assert (a["value"], a["offs"], a["logits"], a["os"], a["ls"]) == (y.plane, b.lin, b.lin + 8 * L * P * 2 * f4,
                                                                            str(N_LIN), str(N_LIN))
            assert (a["ref"], a["vis"], a["order"], a["slots"]) == (b.ref_cam, _addr(28) + b.m0 * 4, b.order, b.slots)
            assert (a["B"], a["NC"], a["S"], a["L"], a["P"], a["Z"]) == ("1", str(NC), str(S), str(L), str(P), str(Z))
            assert by[(l, "A", i)][0] in after[idx]
            # the planes' event was waited for on this stream before the gather
            assert any(o == ("wait", b.stream, 0xABC000) for o in ops[:idx])
            idx, a = by[(l, "B", i)]
            assert (a["a"], a["res"], a["w"], a["bias"]) == (b.slots, b.x1, y.wB, y.biasB)
            assert (a["g1"], a["b1"], a["g2"], a["b2"], a["y"], a["ldy"]) == (y.ln1_g, y.ln1_b, y.ln2_g, y.ln2_b,
                                                                           y.out + b.m0 * 256 * f4, "256")
            if tail:
                assert (a["qterm"], a["zq"], a["zv"], a["nq"], a["ldzq"], a["ldq"]) == (
                    y.q_term + b.m0 * NQ_TAIL * f4, y.zq + b.m0 * NQ_TAIL * f4, y.zv + b.m0 * 256 * f4, str(NQ_TAIL),
                    str(NQ_TAIL), str(NQ_TAIL))
            else:
                assert (a["qterm"], a["zq"], a["zv"], a["nq"]) == (0, 0, 0, "0")
            assert by[(l, "S", i)][0] in after[idx]
    # when the call returns, the caller's stream is ordered behind every launch
    assert last_on.get(MAIN, set()) >= {k[0] for k in kernels}
    # and every band stream started behind the caller's stream (fork): its first op is a wait on an event recorded on MAIN
    for b in bands:
        if b.stream != MAIN:
            first = next(o for o in ops if o[0] in ("wait",) and o[1] == b.stream)
            rec = next(o for o in ops if o[0] == "record" and o[1] == first[2])
            assert rec[2] == MAIN and ops.index(rec) < ops.index(first)
    assert not any(o[0] == "sync" for o in ops)


def test_failing_launch_still_joins_and_reports(harness):
    layers, bands = _setup(3, (0, 20800, 40000))
    rc, ops, _ = _call(harness, layers, bands, fail_at=5)       # the 6th launch: program A... of layer 0, band 1 is #3; #5 = S band 1
    assert rc == -2
    kernels, after, last_on = _happens_before(ops)
    assert len(kernels) == 6                                    # nothing is launched after the failure
    assert last_on.get(MAIN, set()) >= {k[0] for k in kernels}  # but the streams are joined


def test_bands_on_one_stream_need_no_events_between_them(harness):
    layers, bands = _setup(2, (0, 20800, 40000), same_stream=True)
    rc, ops, _ = _call(harness, layers, bands)
    assert rc == 0
    assert [o for o in ops if o[0] == "wait" and o[2] != 0xABC000] == []     # only the planes' event is waited for
