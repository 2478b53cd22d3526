#!/bin/bash
# Round 5: counter evidence for every kernel in the bench line.
#  * HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md; FETCH_SIZE x 2 on gfx950), summed over all half-step-type
#    kernels of tools/pmc_probe.py <config>, per walker-update;
#  * one SQ pass for the kernels without any so far (k_halfstep_slab, k_persist_valu, k_persist LOCAL, k_persist_mix, k_mt_tok).
# -> gpurun_out/r05/pmc/summary.json (copied to profiles/r05/pmc_traffic_r05.json)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05/pmc
mkdir -p $O
STEPS=40
for cfg in c2 c2_store c3 c4 c5 w128 w128_de valu2048 local2048; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${cfg}_$ctr
    ( cd /tmp && timeout 200 rocprofv3 --pmc $ctr -d $O/${cfg}_$ctr -o p -f csv -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py $cfg $STEPS > $O/${cfg}_$ctr.log 2>&1 )
  done
done
for cfg in w128 valu2048 local2048 c4 exact_c3; do
  rm -rf $O/${cfg}_SQ
  ( cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/${cfg}_SQ -o p -f csv -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py $cfg $STEPS > $O/${cfg}_SQ.log 2>&1 )
done
python - "$O" <<'PY'
import collections, csv, glob, json, os, re, sys
O = sys.argv[1]
HOT = re.compile(r"k_halfstep|k_persist|k_wide|k_replay|k_small_run")
out = {"how": "tools/pmc_r05.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of tools/pmc_probe.py <config> 40; bytes = (2 x FETCH_SIZE + "
              "WRITE_SIZE) x 1024 (FETCH_SIZE doubled: gfx950 counts 128-byte requests at 64), summed over the half-step-type kernels of the process "
              "(launched more than twice), divided by walkers x steps", "configs": {}, "sq": {}}
def info(cfg, ctr):
    for ln in open(os.path.join(O, "%s_%s.log" % (cfg, ctr)), errors="replace"):
        if ln.startswith("{") and '"config"' in ln:
            return json.loads(ln)
    return None
for d in sorted(glob.glob(O + "/*_FETCH_SIZE")):
    cfg = os.path.basename(d)[:-len("_FETCH_SIZE")]
    rec = {}
    meta = None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        meta = info(cfg, ctr) or meta
        per = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(O, "%s_%s" % (cfg, ctr), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr and HOT.search(r["Kernel_Name"]):
                    k = re.sub(r"\(.*", "", r["Kernel_Name"])
                    per[k][0] += float(r["Counter_Value"])
                    per[k][1] += 1
        rec[ctr] = {k: {"sum_kb": v[0], "launches": v[1]} for k, v in per.items() if v[1] > 2}
    if not meta or not rec["FETCH_SIZE"]:
        out["configs"][cfg] = {"error": "no samples", "meta": meta}
        continue
    wu = meta["N"] * meta["steps"]
    fetch = sum(v["sum_kb"] for v in rec["FETCH_SIZE"].values())
    write = sum(v["sum_kb"] for v in rec["WRITE_SIZE"].values())
    b = (2 * fetch + write) * 1024 / wu
    out["configs"][cfg] = {"bytes_per_walker_update": b, "fetch_kb_total": fetch, "write_kb_total": write, "walker_updates": wu,
                           "algorithmic_bytes_per_walker_update": meta["algorithmic_bytes_per_walker_update"],
                           "traffic_over_algorithmic": b / meta["algorithmic_bytes_per_walker_update"], "kernels": rec, "probe": meta}
    print("%-10s %8.1f B per walker-update measured, %8.1f algorithmic (x %.2f)  kernels: %s" % (
        cfg, b, meta["algorithmic_bytes_per_walker_update"], b / meta["algorithmic_bytes_per_walker_update"], ", ".join(sorted(rec["FETCH_SIZE"]))[:150]))
for d in sorted(glob.glob(O + "/*_SQ")):
    cfg = os.path.basename(d)[:-3]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per[re.sub(r"\(.*", "", r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out["sq"][cfg] = {}
    for k, cs in per.items():
        n = max(len(v) for v in cs.values())
        if n <= 2 or not re.search(r"k_halfstep|k_persist|k_mt_tok|k_fin|k_mt_gen|k_wide", k):
            continue
        med = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
        med["launches"] = n
        if med.get("SQ_WAVE_CYCLES"):
            med["wait_inst_any_over_wave_cycles"] = med.get("SQ_WAIT_INST_ANY", 0.0) / med["SQ_WAVE_CYCLES"]
            med["mfma_busy_over_busy_cycles"] = med.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, med.get("SQ_BUSY_CYCLES", 1.0))
        out["sq"][cfg][k] = med
        print("SQ %-10s %-60s waves %-8.0f wait/wave-cycles %.2f  valu %.3g  lds %.3g" % (cfg, k[:60], med.get("SQ_WAVES", 0), med.get("wait_inst_any_over_wave_cycles", 0), med.get("SQ_INSTS_VALU", 0), med.get("SQ_INSTS_LDS", 0)))
json.dump(out, open(O + "/summary.json", "w"), indent=1)
PY
find $O -name "*.csv" -size +512k -delete
find $O -name "*.db" -delete
