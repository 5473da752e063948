#!/bin/bash
# FETCH_SIZE calibration on the GPU box: tools/fetch_calib.sh  -> gpurun_out/fetch_calibration.json
R=$PWD; mkdir -p $R/gpurun_out/fetch_calib; export TMPDIR=/tmp; cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/tools/probes/fetch_calib.hip 2>/dev/null || exit 1
timeout -k 5 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/fetch_calib -o pmc --output-format csv -- /tmp/fetch_calib > $R/gpurun_out/fetch_calib/known.txt 2>&1
cd $R; python - <<'PY'
import csv, glob, json
known = {}
for ln in open("gpurun_out/fetch_calib/known.txt"):
    if ln.startswith("KNOWN"):
        t = ln.split(); known[t[1]] = [int(x) for x in t[2:]]
got = {}
for f in glob.glob("gpurun_out/fetch_calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k in known and r["Counter_Name"] == "FETCH_SIZE":
            got[k] = float(r["Counter_Value"]) * 1024.0
out = {}
for k, kb in known.items():
    if k in got:
        out[k] = {"fetch_size_bytes_raw": got[k], "known_bytes": kb, "factor_to_known": [round(b / got[k], 3) for b in kb]}
out["reading"] = ("factor_to_known = known bytes / raw FETCH_SIZE: [0] whole 64-byte units, [1] bytes touched.  calib_stream: the x2 of "
                  "MI355X_MICROARCH.md; gather64 = a 64-byte splat record per lane (compositing kernels' stage loads); gather4 = a 4-byte word "
                  "per lane (the binning's gid_of[slot] gathers)")
json.dump(out, open("gpurun_out/fetch_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
