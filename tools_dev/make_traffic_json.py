#!/usr/bin/env python
"""profiles/sca_gather_traffic.json from rocprofv3 PMC csv files (one counter set per pass, --kernel-trace only):
    python tools_dev/make_traffic_json.py <kernel regex> <out.json> <source note> a_counters.csv b_counters.csv ...
HBM bytes per launch = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE counts 128-byte requests at 64 B,
MI355X_MICROARCH.md §HBM) + WRITE_SIZE x 1024.  The file is keyed on the digest of the kernel's source so bench.py
only trusts it for the code it was measured on."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import build, ext  # noqa: E402

rx, out, note = re.compile(sys.argv[1]), sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: [0, 0.0])
name = None
for path in sys.argv[4:]:
    for row in csv.DictReader(open(path)):
        if rx.search(row['Kernel_Name']):
            name = row['Kernel_Name'].split('(')[0].replace('void ', '')
            a = acc[row['Counter_Name']]
            a[0] += 1
            a[1] += float(row['Counter_Value'])
m = lambda n: acc[n][1] / acc[n][0] if acc[n][0] else None
fetch, write = m('FETCH_SIZE'), m('WRITE_SIZE')
hit, miss = m('TCC_HIT_sum'), m('TCC_MISS_sum')
busy = m('SQ_BUSY_CYCLES')
res = {
    "kernel": name, "kernel_variant": ext.sca_variant_name(), "source_digest": build.source_digest("sca_fused.hip"),
    "hbm_bytes_per_launch": None if fetch is None or write is None else int(2 * fetch * 1024 + write * 1024),
    "hbm_read_bytes_per_launch": None if fetch is None else int(2 * fetch * 1024),
    "hbm_write_bytes_per_launch": None if write is None else int(write * 1024),
    "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write,
    "l1_cache_line_accesses_per_launch": m('TCP_TOTAL_CACHE_ACCESSES_sum'),
    "l1_to_l2_read_requests_per_launch": m('TCP_TCC_READ_REQ_sum'),
    "ta_busy_frac": None if not (m('TA_BUSY_avr') and busy) else m('TA_BUSY_avr') / (busy / 32),
    "l2_hit": None if hit is None else hit / (hit + miss),
    "vmem_read_wave_instructions_per_launch": m('SQ_INSTS_VMEM_RD'),
    "wave_cycles_waiting_frac": None if not m('SQ_WAVE_CYCLES') else m('SQ_WAIT_ANY') / m('SQ_WAVE_CYCLES'),
    "wave_cycles_issue_stalled_frac": None if not m('SQ_WAVE_CYCLES') else m('SQ_WAIT_INST_ANY') / m('SQ_WAVE_CYCLES'),
    "launches_averaged": acc['FETCH_SIZE'][0], "source": note, "round": 6,
}
with open(out, 'w') as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
