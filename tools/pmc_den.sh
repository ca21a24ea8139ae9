#!/bin/bash
# SQ counters / traffic of the kernels of ndwt_denoise (Den3, Fwd3 LOWONLY, ...):  tools/pmc_den.sh <tag> [n wname level]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/pmc_den_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python tools/run_denoise.py "$@" > $out/stats.log 2>&1
f=$(ls $out/stats/*/*kernel_stats.csv | head -1); grep -E '^"Name"|ndwt::' $f | cut -d, -f1-4 | cut -c1-160
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/sq -- python tools/run_denoise.py "$@" > $out/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq2 -- python tools/run_denoise.py "$@" > $out/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python tools/run_denoise.py "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python tools/run_denoise.py "$@" > $out/write.log 2>&1
python - $out <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ndwt" not in k: continue
        name = k.split("<ndwt::")[1].split(">")[0][:44] if "<ndwt::" in k else k[:44]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = collections.defaultdict(float)
for name in sorted(agg):
    print(name, "launches/call", len(next(iter(agg[name].values()))) // 3, {c: round(sum(v) / len(v)) for c, v in sorted(agg[name].items())})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot[c] += sum(agg[name].get(c, [0])) / 3        # per ndwt_denoise call (3 calls per pass)
print("per ndwt_denoise call: FETCH_SIZE %.0f KiB, WRITE_SIZE %.0f KiB, HBM-side bytes (2*FETCH + WRITE) %.2f GB" %
      (tot["FETCH_SIZE"], tot["WRITE_SIZE"], (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / 1e9))
PY
