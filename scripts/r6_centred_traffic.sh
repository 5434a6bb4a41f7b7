#!/bin/bash
# round 6: HBM traffic of the centred collect launch (scan8_kernel<3, 1, 1, true> on enc_like_10M), separate --pmc passes with
# --kernel-trace only; FETCH_SIZE in KB x 1024 x 2 (the guide's gfx950 correction), WRITE_SIZE in KB x 1024
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ct_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/ct_$C -- python $ROOT/scripts/gpu_enc_like.py 10000000 8 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, json
out = {"kernel": "mx::scan8_kernel<3, 1, 1, true>", "algorithmic_bytes_per_launch": 10_000_000 * 384 + 10_000_000 * 4 + 156_250 * 32,
       "algorithmic_note": "int8 rows + 4 bytes of a_c per row + 32 bytes of scales per 64-row tile"}
for c, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    fs = glob.glob(f"/tmp/ct_{c}/**/*_counter_collection.csv", recursive=True)
    vals = {}
    for r in csv.DictReader(open(fs[0])) if fs else []:
        if "scan8_kernel<3, 1, 1, true>" in r["Kernel_Name"] and r["Counter_Name"] == c:
            vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    out[c + "_KB_avg"] = sum(vals.values()) / max(1, len(vals))
    out[c.lower() + "_bytes"] = out[c + "_KB_avg"] * scale
    out["launches_" + c.lower()] = len(vals)
out["traffic_bytes_per_launch"] = out["fetch_size_bytes"] + out["write_size_bytes"]
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
print(json.dumps(out, indent=1))
PY
