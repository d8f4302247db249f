cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/clk
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/clk -- python3 tools/conv_bench.py --iters 3 > gpurun_out/clk/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/clk/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/clk/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:50], r.get("Grid_Size"))
seen = {}
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
        ns, name, grid = dur[r["Dispatch_Id"]]
        if ns > 300000:
            ghz = float(r["Counter_Value"]) / 8.0 / ns
            print(f"{name:52s} grid {grid:>9s}  {ns/1e6:7.3f} ms  effective clock {ghz:5.2f} GHz")
PY
