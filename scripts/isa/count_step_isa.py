#!/usr/bin/env python3
"""Per-solver-step instruction counts of a k_sample_bf16 build, from its device assembly.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -DB16_VARIANT=0 -DB16_EXTRA=0 -DB16_GUIDE=0 -S --cuda-device-only \
          -o k.s generativerl_preview_amd/csrc/engine_bf16_inst.hip
    python scripts/isa/count_step_isa.py k.s > profiles/rNN_isa_counts.json

The step loop is the largest basic block of the kernel (fully unrolled net evaluation).  Issue model (one wave per
SIMD, MI355X_MICROARCH.md "vector-instruction ISSUE cost"): every instruction takes a 4-cycle issue slot, the
transcendentals and the MFMA's own issue 8; the step cannot run faster than max(32 x MFMAs, sum of issue costs)."""
import collections
import json
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
blocks, cur = [], []
for l in lines:
    s = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", s) or re.match(r"^s_cbranch|^s_branch|^s_endpgm", s):
        if cur:
            blocks.append(cur)
        cur = []
        continue
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    cur.append(s)
if cur:
    blocks.append(cur)
body = max(blocks, key=len)
cnt = collections.Counter(x.split()[0] for x in body)
mfma = sum(v for k, v in cnt.items() if k.startswith("v_mfma"))
trans = sum(v for k, v in cnt.items() if re.match(r"v_(exp|rcp|log|rsq|sqrt|sin|cos)_", k))
slots = sum(cnt.values()) + trans + mfma          # 8-cycle instructions take two 4-cycle slots
groups = {
    "mfma": mfma, "transcendental (v_exp/v_rcp)": trans,
    "other VALU": sum(v for k, v in cnt.items() if k.startswith("v_") and not k.startswith("v_mfma")) - trans,
    "v_accvgpr_read/write": sum(v for k, v in cnt.items() if k.startswith("v_accvgpr")),
    "ds_read": sum(v for k, v in cnt.items() if k.startswith("ds_read")),
    "ds_write": sum(v for k, v in cnt.items() if k.startswith("ds_write")),
    "LDS-DMA (buffer_load ... lds)": cnt.get("buffer_load_dwordx4", 0),
    "s_waitcnt": cnt.get("s_waitcnt", 0), "s_nop": cnt.get("s_nop", 0), "s_barrier": cnt.get("s_barrier", 0),
    "scratch": sum(v for k, v in cnt.items() if k.startswith("scratch_")),
    "other SALU": sum(v for k, v in cnt.items() if k.startswith("s_") and k not in ("s_waitcnt", "s_nop", "s_barrier")),
}
print(json.dumps({
    "source": sys.argv[1], "instructions_per_step": sum(cnt.values()), "groups": groups,
    "instructions_per_mfma": sum(cnt.values()) / mfma,
    "issue_cycles_per_step_lower_bound": 4 * slots, "issue_cycles_per_mfma": 4.0 * slots / mfma,
    "mfma_cycles_per_step": 32 * mfma, "by_opcode": dict(cnt.most_common())}, indent=1))
