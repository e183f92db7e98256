"""Lab (round 5): libocc_amd with the gathers' sampling set-up WITHOUT scalar lane masks.

Copies occnet_amd/csrc to a scratch directory, replaces the body of common.h::bilinear_setup_b by the mask-free form of
tools_dev/hazard_micro.hip::bilinear_setup_vb (same weights, offsets and in-map corner count, bit for bit), compiles every
.hip with the product's flags and links tools_dev/bin/libocc_amd_maskfree.so (git-ignored, travels to the GPU box).
The shipped library is not touched.     usage: python tools_dev/lab/maskfree_setup/build_variant.py"""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from occnet_amd import build as B                                  # noqa: E402

OLD_HEAD = "__device__ __forceinline__ int bilinear_setup_b(float loc_x, float loc_y, float attn, int H, int W, int lvl_pix0,"
NEW_BODY = r'''__device__ __forceinline__ int bilinear_setup_b(float loc_x, float loc_y, float attn, int H, int W, int lvl_pix0,
                                                unsigned pix_bytes, unsigned dead, bool live, SampleParamB& sp) {
  // every condition becomes a 0 / 1 VGPR at once (v_cmp + v_cndmask), conditions are combined by VALU integer ANDs and the
  // selects compare those integers: no s_and_b64 / s_and_saveexec_b64 ever reads a lane mask a v_cmp has just written
  auto flag = [](bool c) { int f = c ? 1 : 0; asm volatile("" : "+v"(f)); return f; };
  const float h_im = loc_y * (float)H - 0.5f;
  const float w_im = loc_x * (float)W - 0.5f;
  const int adm = flag(live) & flag(h_im > -1.f) & flag(w_im > -1.f) & flag(h_im < (float)H) & flag(w_im < (float)W);
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = (int)hf, w_low = (int)wf;
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - hf, lw = w_im - wf;
  const float hh = 1.f - lh, hw = 1.f - lw;
  const int t = flag(h_low >= 0) & adm, b = flag(h_high <= H - 1) & adm, l = flag(w_low >= 0), r = flag(w_high <= W - 1);
  const int base = lvl_pix0 + h_low * W + w_low;
  const int c0 = t & l, c1 = t & r, c2 = b & l, c3 = b & r;
  sp.w[0] = c0 ? hh * hw * attn : 0.f; sp.o[0] = c0 ? (unsigned)base * pix_bytes : dead;
  sp.w[1] = c1 ? hh * lw * attn : 0.f; sp.o[1] = c1 ? (unsigned)(base + 1) * pix_bytes : dead;
  sp.w[2] = c2 ? lh * hw * attn : 0.f; sp.o[2] = c2 ? (unsigned)(base + W) * pix_bytes : dead;
  sp.w[3] = c3 ? lh * lw * attn : 0.f; sp.o[3] = c3 ? (unsigned)(base + W + 1) * pix_bytes : dead;
  return c0 + c1 + c2 + c3;
}
'''


def main():
    out = os.path.join(ROOT, "tools_dev", "bin", "libocc_amd_maskfree.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "pkg", "csrc")                # common.h includes "../../include/occnet_amd.h"
        shutil.copytree(B.CSRC, src)
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
        p = os.path.join(src, "common.h")
        s = open(p).read()
        i = s.index(OLD_HEAD)
        j = s.index("\n}\n", i) + 3
        assert "return n_in;" in s[i:j]
        open(p, "w").write(s[:i] + NEW_BODY + s[j:])
        hips = sorted(f for f in os.listdir(src) if f.endswith(".hip"))

        def cc(f):
            o = os.path.join(tmp, f[:-4] + ".o")
            r = subprocess.run([B.hipcc()] + B.FLAGS + ["-I" + os.path.join(ROOT, "include"), "-c", "-o", o, os.path.join(src, f)],
                               capture_output=True, text=True)
            if r.returncode:
                sys.stderr.write(r.stdout + r.stderr)
                raise SystemExit(f"hipcc failed on {f}")
            return o
        with ThreadPoolExecutor(max_workers=8) as ex:
            objs = list(ex.map(cc, hips))
        r = subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout + r.stderr)
            raise SystemExit("link failed")
    print(out)


if __name__ == "__main__":
    main()
