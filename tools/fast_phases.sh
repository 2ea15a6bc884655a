#!/bin/bash
# Per-phase instruction and time budget of k_fast_cells (VERDICT r03 task 2): the kernel's timing probes (FT_DEBUG_FAST) cut it
# off behind successive phases; per probe one rocprofv3 --pmc pass (SQ instruction counters per wave) and the HIP-event time
# per launch of 128 frames when the extractor runs alone.  Object scenes and dense mosaics.  -> gpurun_out/<tag>/fast_phases.json
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-fast_phases}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mosaic in 0 10; do for d in 4 16 3 2 0; do
  export FT_DEBUG_FAST=$d FP_MOSAIC=$mosaic
  python3 $REPO/tools/fast_probe.py > $OUT/time_${mosaic}_$d.txt 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS \
     -d $OUT/pmc_${mosaic}_$d -o pmc -- python3 $REPO/tools/fast_probe.py > $OUT/pmc_${mosaic}_$d.log 2>&1
done; done
unset FT_DEBUG_FAST FP_MOSAIC
python3 - <<PY
import glob, json, re, sqlite3
out = {"method": "k_fast_cells<48,false> cut off behind successive phases by FT_DEBUG_FAST ("
       "see below), 256 resident 1280x720 frames, extractor alone on the chip; per probe the SQ counters per wave (rocprofv3 --pmc, "
       "own pass) and the HIP-event time per launch of 128 frames; a phase = the difference between two probes",
       "probes": {"4": "staging of the tile (HBM -> LDS) and the score-plane clear", "16": "+ phase A: rejection on the compass pairs, verdict masks, ring append (candidates dropped)",
                  "3": "+ phase B without the score network: candidate gather (17 LDS byte reads + packing per candidate pair), three-pixel hash, score store, corner list",
                  "2": "+ the min/max score network (cornerScore)", "0": "+ NMS, threshold fallback, emission = the whole kernel"}, "scenes": {}}
for mosaic, name in ((0, "objects"), (10, "dense_mosaic10")):
    rows = {}
    for d in (4, 16, 3, 2, 0):
        t = open(f"$OUT/time_{mosaic}_{d}.txt").read()
        m = re.search(r"'kernel.fast_cells': ([0-9.]+)", t)
        rec = {"ms_per_launch_alone": float(m.group(1)) if m else None}
        for f in glob.glob(f"$OUT/pmc_{mosaic}_{d}/**/*.db", recursive=True):
            db = sqlite3.connect(f)
            c = {}
            for n, cn, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
                if "k_fast_cells" in n: c[cn] = v
            w = c.get("SQ_WAVES", 0) or 1
            rec.update({"valu_per_wave": round(c.get("SQ_INSTS_VALU", 0) / w, 1), "salu_per_wave": round(c.get("SQ_INSTS_SALU", 0) / w, 1),
                        "lds_per_wave": round(c.get("SQ_INSTS_LDS", 0) / w, 1), "quad_cycles_per_wave": round(c.get("SQ_WAVE_CYCLES", 0) / w),
                        "lds_bank_conflict_over_active_lds": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_ACTIVE_INST_LDS", 0), 1), 3),
                        "wait_inst_lds_frac_of_wave_cycles": round(c.get("SQ_WAIT_INST_LDS", 0) / max(c.get("SQ_WAVE_CYCLES", 0), 1), 3)})
        rows[str(d)] = rec
    ph = {}
    order = [("staging", None, "4"), ("rejection + ring append (A)", "4", "16"), ("candidate gather + packing + corner list (B minus network)", "16", "3"),
             ("score network", "3", "2"), ("NMS + fallback + emission", "2", "0")]
    for name2, a, b in order:
        ph[name2] = {k: round(rows[b][k] - (rows[a][k] if a else 0), 4) for k in ("valu_per_wave", "salu_per_wave", "lds_per_wave", "ms_per_launch_alone") if rows[b].get(k) is not None}
    out["scenes"][name] = {"probes": rows, "phases": ph}
json.dump(out, open("$OUT/fast_phases.json", "w"), indent=1)
print(json.dumps({k: v["phases"] for k, v in out["scenes"].items()}, indent=1))
PY
rm -rf $OUT/pmc_*_*/
