"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A restatement of the reference's CPU path for the OccNet / BEVFormer-occ forward hot path.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(occnet_amd/) never does.

Pinning status: the reference tree holds no tests, golden vectors or fixtures for this path
 and its arithmetic kernel lives in un-vendored mmcv-full
(mmcv.ops.multi_scale_deform_attn.multi_scale_deformable_attn_pytorch, pinned only by a hyperlink to
BEVFormer's install guide: mmcv-full 1.4.0).  What IS pinned here:
  * oracle/msda.py restates that mmcv function from its published algorithm and is cross-checked
    against an independent fp64 scalar re-derivation of mmcv's CUDA kernel arithmetic
    (tests/test_oracle_msda.py) — "parity unpinned" for this one function;
  * the module-level oracle (oracle/model.py) is checked against golden vectors produced by running
    the reference's OWN module files from /root/reference under thin mmcv stubs
    (oracle/refshim/, generator oracle/gen_golden.py, fixtures tests/golden/*.npz).
"""
