#!/usr/bin/env python3
"""Opcode-weighted issue cost of the hot kernels' vector instruction streams.

Until round 5 every "vector-issue bound" in this repo was  instructions x 4 cycles / (1 024 SIMDs x 2.4 GHz).  The issue-cost table
measured in round 6 (profiles/r06_valu_rates.txt, tools/valu_rate.hip) has classes: ~2.3 (v_mov, 16-bit VOP2, f32 add / mul),
~2.9 (32-bit integer add / sub / logic / shift right, f32 fma, v_bitop3), ~4.3 (the rest), ~8.3 (v_rcp, v_sqrt, v_max3_u16, ...).
This tool prices a kernel's stream with it:

  1. the gfx950 ISA of the kernel (hipcc -S of the csrc file, the Makefile's flags), split into basic blocks;
  2. a weight per block: LOOP_TRIPS ** (number of loops the block sits in) - loops = backward branches; the dynamic counts of a
     kernel come from its loop bodies, the straight-line prologue runs once.  An estimate of the dynamic mix, not a trace: the
     SQ counters give the dynamic instruction COUNT (SQ_INSTS_VALU), this gives the mean cycles per instruction to multiply it by;
  3. every vector instruction priced by the table (an SGPR / vcc source operand, SDWA and DPP forms are dear whatever the opcode -
     measured); opcodes the table does not hold are priced at the dear class and listed.

usage: tools/valu_mix.py [--rates profiles/r06_valu_rates.txt] [--out profiles/r06_valu_mix.json]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fasttrack_amd", "csrc")
LOOP_TRIPS = 6
DEAR = 4.33

# (file, symbol fragment, name in the profiles)
KERNELS = [
    ("kernels_extract.hip", "k_fast_cellsILi48ELb0EE", "k_fast_cells<48, false>"),
    ("kernels_extract.hip", "k_fast_cellsILi64ELb0EE", "k_fast_cells<64, false>"),
    ("kernels_extract.hip", "k_orient_descILi2EE", "k_orient_desc<2>"),
    ("kernels_extract.hip", "k_pyr_rowsILb0EE", "k_pyr_rows<false>"),
    ("kernels_extract.hip", "k_pyr_rowsILb1EE", "k_pyr_rows<true>"),
    ("kernels_extract.hip", "9k_compactE", "k_compact"),
    ("kernels_octree.hip", "8k_octreeE", "k_octree"),
    ("kernels_match.hip", "14k_stereo_matchE", "k_stereo_match"),
    ("kernels_search.hip", "19k_fisheye_2nn_batchE", "k_fisheye_2nn_batch"),
    ("kernels_search.hip", "15k_resolve_batchILb0ELb1EE", "k_resolve_batch<false>"),
    ("kernels_search.hip", "15k_resolve_batchILb1ELb1EE", "k_resolve_batch<true>"),
    ("kernels_search.hip", "19k_search_last_firstE", "k_search_last_first"),
    ("kernels_search.hip", "20k_search_local_firstE", "k_search_local_first"),
]


def load_rates(path):
    """label -> cycles at 2.4 GHz by events, W = 8 rows"""
    t = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+W=8\s+[\d.]+ cycles \(s_memtime\)\s+([\d.]+) at 2.4 GHz", line)
        if m:
            t[m.group(1).strip()] = float(m.group(2))
    return t


def isa_of(src):
    flags = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc"
    out = f"/tmp/valu_mix_{os.path.basename(src)}.s"
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        cmd = f"/opt/rocm/bin/hipcc --offload-arch=gfx950 {flags} -I{ROOT}/include -S --cuda-device-only -o {out} {src}"
        subprocess.run(cmd, shell=True, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def blocks_of(lines, frag):
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(frag) + r"\w*:", l)]
    if not starts:
        return None
    s = starts[0]
    e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": "entry", "insts": [], "branches": []}
    for l in lines[s + 1:e]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                blocks.append(cur)
                cur = {"label": m.group(1), "insts": [], "branches": []}
            continue
        op = t.split()[0]
        cur["insts"].append(t.split(";")[0].strip())
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["branches"].append(t.split()[1])
    blocks.append(cur)
    return blocks


def loop_depths(blocks):
    index = {b["label"]: i for i, b in enumerate(blocks)}
    depth = [0] * len(blocks)
    for j, b in enumerate(blocks):
        for tgt in b["branches"]:
            i = index.get(tgt)
            if i is not None and i <= j:  # backward branch: blocks i .. j form a loop (the compiler lays loops out contiguously)
                for k in range(i, j + 1):
                    depth[k] += 1
    return depth


FAST_SRC_SENSITIVE = {"v_lshrrev_b32 const", "v_ashrrev_i32 const", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32",
                      "v_ashrrev_i32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_fma_f32", "v_bitop3_b32",
                      "v_max_u16", "v_min_u16", "v_add_u16", "v_sub_u16", "v_max_i16", "v_lshlrev_b16", "v_lshrrev_b16", "v_mul_lo_u16"}
ALIAS = {"v_subrev_f32": "v_sub_f32", "v_min_i16": "v_max_i16", "v_lshlrev_b32": "v_lshlrev_b32 const", "v_lshrrev_b32": "v_lshrrev_b32 const",
         "v_ashrrev_i32": "v_ashrrev_i32 const", "v_addc_co_u32": "v_addc_co_u32 vcc", "v_add_co_u32": "v_add_co_u32 vcc",
         "v_sub_co_u32": "v_sub_co_u32 vcc", "v_subb_co_u32": "v_addc_co_u32 vcc", "v_subrev_co_u32": "v_sub_co_u32 vcc",
         "v_subbrev_co_u32": "v_addc_co_u32 vcc", "v_cvt_u32_f32": "v_cvt_i32_f32", "v_ceil_f32": "v_floor_f32", "v_max3_i32": "v_max3_u32",
         "v_min3_i32": "v_min3_u32", "v_cvt_f64_f32": "v_add_f64", "v_cvt_f64_i32": "v_add_f64", "v_cvt_f64_u32": "v_add_f64",
         "v_cvt_f32_f64": "v_add_f64", "v_cvt_i32_f64": "v_add_f64", "v_fmac_f64": "v_fma_f64", "v_div_scale_f32": "v_fma_f32 dear",
         "v_ldexp_f32": "v_fma_f32 dear", "v_div_fmas_f32": "v_fma_f32 dear", "v_div_fixup_f32": "v_fma_f32 dear"}


def price(inst, rates, unknown):
    parts = inst.replace(",", " ").split()
    op = parts[0]
    if not op.startswith("v_"):
        return None
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    form = op[len(base) + 1:] if len(op) > len(base) else ""
    if form in ("sdwa", "dpp") or "row_" in inst or "quad_perm" in inst or "dst_sel" in inst:
        return rates.get(f"{base} {form}".strip(), DEAR), "dear"
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        return rates.get("v_cmp_gt_u32 vcc", 4.4), "dear"
    if base == "v_cndmask_b32":
        return rates.get("v_cndmask_b32 e64 sgpr", DEAR), "dear"
    b = ALIAS.get(base, base)
    if b in rates:
        c = rates[b]
        if b in FAST_SRC_SENSITIVE:
            srcs = parts[2:]
            if any(re.match(r"^(s\d+|s\[|vcc|exec|ttmp|m0)", x) for x in srcs):
                return rates.get("v_and_b32 sgpr", DEAR), "dear"   # a scalar-register source makes a cheap opcode dear (measured)
            if form == "e64" and f"{b} e64" in rates:
                c = rates[f"{b} e64"]
            if b == "v_mov_b32" and form == "e64":
                return rates.get("v_mov_b32 e64", DEAR), "dear"
        cls = "cheap" if c < 3.4 else ("dear" if c < 6.0 else "slow8")
        return c, cls
    # not in the table: by family
    if re.match(r"v_(rcp|rsq|sqrt|log|exp|sin|cos)", base):
        return rates.get("v_rcp_f32", 8.2), "slow8"
    if base.endswith("_f64") or base.endswith("_u64") or base.endswith("_i64") or base.endswith("_b64"):
        return rates.get("v_fma_f64", 5.2) if "fma" in base or "mul" in base else rates.get("v_add_f64", 4.6), "dear"
    unknown[base] += 1
    return DEAR, "dear"


def mix_of(blocks, rates):
    depth = loop_depths(blocks)
    unknown = collections.Counter()
    tot_w = tot_c = 0.0
    static = 0
    cls_w = collections.Counter()
    op_w = collections.Counter()
    for b, d in zip(blocks, depth):
        w = float(LOOP_TRIPS ** min(d, 4))
        for inst in b["insts"]:
            p = price(inst, rates, unknown)
            if p is None:
                continue
            c, cls = p
            static += 1
            tot_w += w
            tot_c += w * c
            cls_w[cls] += w
            op_w[re.sub(r"_(e32|e64)$", "", inst.split()[0])] += w
    if tot_w == 0:
        return None
    top = [(k, round(v / tot_w, 3)) for k, v in op_w.most_common(8)]
    return {"static_valu": static, "mean_cycles_per_valu": round(tot_c / tot_w, 3),
            "share_by_class": {k: round(v / tot_w, 3) for k, v in sorted(cls_w.items())},
            "top_opcodes_weighted": top, "priced_at_dear_class_unmeasured": dict(unknown)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rates", default=os.path.join(ROOT, "profiles", "r06_valu_rates.txt"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_valu_mix.json"))
    a = ap.parse_args()
    rates = load_rates(a.rates)
    out = {"method": __doc__.split("\n\n")[1].strip().replace("\n", " "),
           "rates": os.path.relpath(a.rates, ROOT), "loop_trips_assumed": LOOP_TRIPS, "clock_hz_nominal": 2.4e9, "simds": 1024,
           "classes": {"cheap": "< 3.4 cycles at 2.4 GHz", "dear": "3.4 - 6", "slow8": ">= 6"}, "kernels": {}}
    cache = {}
    for f, frag, name in KERNELS:
        if f not in cache:
            cache[f] = isa_of(os.path.join(CSRC, f))
        bl = blocks_of(cache[f], frag)
        if bl is None:
            print("not found:", frag, file=sys.stderr)
            continue
        m = mix_of(bl, rates)
        if m:
            out["kernels"][name] = m
            print(f"{name:28s} static {m['static_valu']:5d}  mean {m['mean_cycles_per_valu']:.2f} cycles/VALU  {m['share_by_class']}")
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
