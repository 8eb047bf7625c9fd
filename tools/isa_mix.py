#!/usr/bin/env python3
"""Instruction-class accounting of k_integrate's hot path: the per-task instruction mix of the compiled kernel
(gfx950 ISA of the <shared camera, plain weights> instantiation) priced with the issue rates MEASURED by
tools/ubench/ubench_valu (profiles/r02a_ubench_valu*.log), against the cycles a task really takes.

usage: python tools/isa_mix.py [kernel.s]      (default: compiles dynslam_amd/csrc/dsr_engine.hip to ISA)
The hot path is found structurally: basic blocks of the task loop, minus the blocks that contain the IEEE division
sequence (camera-plane fallback, colour pass: rare) — the colour pass is added back with its measured frequency."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = {"v_fma_f32", "v_mul_f32_e32", "v_mul_f32_e64", "v_add_f32_e32", "v_add_f32_e64", "v_sub_f32_e32", "v_subrev_f32_e32",
        "v_fmac_f32_e32", "v_fmamk_f32", "v_fmaak_f32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_mov_b32_e32",
        "v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_ashrrev_i32_e32", "v_not_b32_e32"}
RATE = {"full": 2.6, "full+sgpr": 4.2, "half": 4.25, "packed": 5.2, "rcp": 8.6, "wide": 8.5}  # cycles per wave64 instruction per SIMD


def classify(ins):
    op = ins.split()[0]
    rest = ins[len(op):]
    if op.startswith("v_pk_"):
        return "packed"
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq")):
        return "rcp"
    if op in ("v_mad_u64_u32", "v_lshl_add_u64", "v_lshlrev_b64", "v_mov_b64_e32") or "_f64" in op:
        return "wide" if op != "v_mov_b64_e32" else "packed"
    if op in FULL:
        return "full+sgpr" if re.search(r"[ ,\-|]s\[?\d", rest) or "vcc" in rest or "exec" in rest else "full"
    return "half"


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        src = os.path.join(ROOT, "dynslam_amd", "csrc", "dsr_engine.hip")
        sys.path.insert(0, ROOT)
        from __graft_entry__ import HIPCC_FLAGS  # the flags the library is built with (minus the link step)
        flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
        text = subprocess.check_output(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", "-", src],
                                       stderr=subprocess.DEVNULL).decode()
    # the instantiation the engine launches: <shared camera, plain weights, 8 voxels per lane, 7 waves, XLDS> (env ISA_MIX_XLDS=0: the round-3 form)
    xl = "Lb0" if os.environ.get("ISA_MIX_XLDS") == "0" else "Lb1"
    m = re.search(r"^(_ZN3dsr11k_integrateILb1ELb1ELi8ELi7E" + xl + r"EE\w*):.*?s_endpgm", text, re.S | re.M)
    body = m.group(0).split("\n")
    blocks, cur = [], ["entry", []]
    for l in body:
        lab = re.match(r"^(\.LBB\d+_\d+):", l)
        if lab:
            blocks.append(cur); cur = [lab.group(1), []]
        elif re.match(r"\s+(v_|s_|ds_|global_|buffer_)", l):
            cur[1].append(l.strip())
    blocks.append(cur)
    # the task loop: from the first block that prefetches a hash entry inside the loop to the plane store
    names = [b[0] for b in blocks]
    first_gather = next(i for i, b in enumerate(blocks) if any("buffer_load_dword" in x for x in b[1]))
    store = next(i for i, b in enumerate(blocks) if any("global_store_dwordx4" in x for x in b[1]))
    hot, pend, colour = [], [], []
    for i in range(first_gather - 4, store + 2):
        b = blocks[i]
        if any("v_div_scale" in x for x in b[1]):
            continue  # IEEE division: camera-plane fallback (rare)
        if any("ds_write_b32" in x for x in b[1]) and sum(x.startswith("v_") for x in b[1]) < 12:
            pend.append(b)  # append to the pending colour list: only in tasks with colour voxels
        else:
            hot.append(b)
    colour = [b for b in blocks if any("global_load_ubyte" in x or "global_store_byte" in x or ("v_div_scale" in x) for x in b[1])
              and sum(x.startswith("v_") for x in b[1]) > 150][:1]
    # one of the two copies of phase A2 runs per task (the compiler versions the loop on a uniform flag): halve that region
    a2 = [b for b in hot if blocks.index(b) > first_gather + 9 and not any("buffer_load" in x for x in b[1])]
    weights = {id(b): 1.0 for b in hot}
    dup = [b for b in a2 if sum(x.startswith("v_") for x in b[1]) >= 20]
    if len(dup) >= 14:  # both versions present
        for b in a2:
            weights[id(b)] = 0.5
    tot = {k: 0.0 for k in RATE}
    salu = 0.0
    half_ops = {}
    for grp, w in ((hot, None), (pend, 0.35 * 0.5), (colour, 10.2 / 64.0)):
        for b in grp:
            ww = weights.get(id(b), 1.0) if w is None else w
            for ins in b[1]:
                if ins.startswith("v_"):
                    k = classify(ins)
                    tot[k] += ww
                    if k == "half":
                        half_ops[ins.split()[0]] = half_ops.get(ins.split()[0], 0.0) + ww
                elif ins.startswith("s_") and not ins.startswith(("s_waitcnt", "s_nop")):
                    salu += ww
    n = sum(tot.values())
    cyc = sum(tot[k] * RATE[k] for k in tot)
    print(f"hot path of k_integrate<shared camera, plain weights>: {len(hot)} blocks + {len(pend)} colour-append blocks (x0.35) "
          f"+ colour pass (x{10.2 / 64:.2f})")
    for k in tot:
        print(f"  {k:10s} {tot[k]:7.1f} instructions/task  x {RATE[k]:4.2f} cycles = {tot[k] * RATE[k]:7.0f}")
    print(f"  VALU total {n:7.1f} per task (SQ_INSTS_VALU / visible blocks measured: 614), weighted mean {cyc / n:.2f} cycles/instruction")
    print(f"  SALU       {salu:7.1f} per task x 4.6 cycles = {salu * 4.6:.0f} (own issue port: overlaps VALU of other waves)")
    groups = {"compares": ("v_cmp",), "selects": ("v_cndmask",), "conversions": ("v_cvt",), "SGPR spills to VGPR lanes": ("v_readlane", "v_writelane"),
              "shifts / bit-field / permute": ("v_lsh", "v_bf", "v_perm", "v_and_or", "v_or3"), "min / max": ("v_min", "v_max", "v_med3"),
              "integer multiply-add": ("v_mad", "v_mul_lo", "v_mul_u32", "v_mul_i32"), "lane counting (colour list)": ("v_mbcnt", "v_readfirstlane")}
    print("  half-rate instructions by kind (per task):")
    rest = dict(half_ops)
    for g, pre in groups.items():
        n_g = sum(v for k, v in half_ops.items() if k.startswith(pre))
        for k in [k for k in rest if k.startswith(pre)]:
            rest.pop(k)
        print(f"    {g:32s} {n_g:6.1f}  = {n_g * RATE['half']:5.0f} cycles")
    print(f"    {'other':32s} {sum(rest.values()):6.1f}")
    measured = 608e-6 * 2.3e9 * 1024 / 616948
    print(f"  VALU issue {cyc:.0f} cycles per task vs {measured:.0f} SIMD-cycles per task measured (608 us x 2.3 GHz x 1024 SIMDs / 616948 "
          f"tasks): {100 * cyc / measured:.0f} %")


if __name__ == "__main__":
    main()
