# SQ counters of the patch kernels on one tower layer (scratch/fp8_micro.py): LDS conflicts / stalls vs MFMA busy
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_patch
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/a -o p --output-format csv -- python $ROOT/scratch/fp8_micro.py 8 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/b -o p --output-format csv -- python $ROOT/scratch/fp8_micro.py 8 > /dev/null 2>&1
python - <<PY
import csv, collections, glob
for sub in ("a", "b"):
    rows = []
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        rows += list(csv.DictReader(open(f)))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        n = r["Kernel_Name"]
        if "patch" not in n: continue
        k = "fp8" if "fp8" in n else ("bf16_gn" if "<true>" in n or "Lb1" in n else "bf16")
        acc[k + " " + n[-40:-20]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(k, {cn: "%.3g" % (sum(v) / len(v)) for cn, v in c.items()})
PY
