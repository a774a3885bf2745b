# Per-width calibration of FETCH_SIZE / WRITE_SIZE (KiB counters) on the GPU box: bash tests/fetch_calibration.sh [out.md]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/r03_fetch_calibration.md}
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/tests/cpp/fetch_calibration_probe.cpp -o /tmp/fetch_cal || exit 1
rm -rf /tmp/fcal; mkdir -p /tmp/fcal
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fcal/$c -- /tmp/fetch_cal > /tmp/fcal/$c.log 2>&1 || echo "pass $c failed"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
cal = dict(zip(*[iter(open("/tmp/fcal/FETCH_SIZE.log").read().split("CAL ")[1].split())] * 2))
cal = {k: int(v) for k, v in cal.items()}
val = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("/tmp/fcal/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        val[k][c] = sum(v) / len(v)
B = cal["bytes_stream"]
rows = [("k_read16", "16-byte loads (double2 value stream)", B, "FETCH_SIZE"), ("k_read4", "4-byte loads (ushort2 column stream)", B, "FETCH_SIZE"),
        ("k_read8", "8-byte loads, coalesced", B, "FETCH_SIZE"), ("k_write8", "8-byte stores, coalesced", B, "WRITE_SIZE")]
L = ["# FETCH_SIZE / WRITE_SIZE calibration on gfx950 (tests/fetch_calibration.sh, tests/cpp/fetch_calibration_probe.cpp)", "",
     "Every kernel touches a known number of distinct bytes exactly once (1 GiB per stream, grid 2048 x 256); counters are in KiB.", "",
     "| kernel | access | bytes touched | counter | counter x 1024 | bytes / (counter x 1024) |", "|---|---|---|---|---|---|"]
for k, what, b, c in rows:
    v = val.get(k, {}).get(c)
    if v is None: continue
    L.append("| `%s` | %s | %d | %s | %.4g | **%.3f** |" % (k, what, b, c, v * 1024, b / (v * 1024)))
g = val.get("k_gather8", {}).get("FETCH_SIZE")
if g is not None:
    idxb = cal["gather_index_bytes"]
    L += ["", "Sorted gather (`k_gather8`, %d indices with gaps of 2..5 entries): FETCH_SIZE x 1024 = %.4g.  The index list itself is a 4-byte stream of %d B;"
          " useful x bytes %d, distinct 64-byte lines %d B, distinct 128-byte lines %d B." % (cal["gather_indices"], g * 1024, idxb, cal["gather_useful_bytes"],
          cal["gather_lines64_bytes"], cal["gather_lines128_bytes"]), ""]
    f4 = B / (val["k_read4"]["FETCH_SIZE"] * 1024) if "k_read4" in val else 2.0
    rest = g * 1024 - idxb / f4
    L.append("After taking out the index stream at the 4-byte factor, the gather accounts for %.4g counter bytes: x %.3f = its 64-byte-line footprint, x %.3f = its "
             "128-byte-line footprint." % (rest, cal["gather_lines64_bytes"] / rest, cal["gather_lines128_bytes"] / rest))
open(out, "w").write("\n".join(L) + "\n")
print("\n".join(L))
PY
