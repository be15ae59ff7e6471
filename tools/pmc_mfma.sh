# MFMA utilisation per kernel of the bench step (north_star: "MFMA-utilisation counters against gfx950 peak").
# PMC only with --kernel-trace (MI355X_MICROARCH.md); one pass, four counters.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rooflines --no-other-configs > /tmp/pmc_mfma.log 2>&1
f=$(find /tmp/pmc_mfma -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY' > $R/gpurun_out/mfma_util.txt
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    k = re.sub(r'\(.*', '', k)[:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
print('# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rooflines --no-other-configs')
print('# per-dispatch averages. MFMA busy is summed over the 1024 SIMDs (32 cycles per v_mfma_f32_16x16x4_f32); GRBM_GUI_ACTIVE is summed over the 8 XCDs.')
print('# mfma_util = MFMA_BUSY / (1024 * GRBM_GUI_ACTIVE / 8): fraction of all matrix pipes busy while the kernel runs (kernels run slower under the counter pass).')
print('%-62s %6s %14s %14s %10s' % ('kernel', 'calls', 'mfma_busy', 'gui_active/8', 'mfma_util'))
rows = []
for k, c in acc.items():
    n = cnt[k]
    mb, ga = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / n, c.get('GRBM_GUI_ACTIVE', 0) / n / 8
    rows.append((mb * n, k, n, mb, ga))
for _, k, n, mb, ga in sorted(rows, reverse=True)[:16]:
    print('%-62s %6d %14.0f %14.0f %9.1f%%' % (k, n, mb, ga, 100 * mb / (1024 * ga) if ga else 0))
PY
cat $R/gpurun_out/mfma_util.txt
