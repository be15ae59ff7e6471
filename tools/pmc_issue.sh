# Where the waves' cycles go, per kernel of a training step: parked (s_waitcnt / barrier), stalled at issue, or issuing --
# and the instruction mix per wave.  Two SQ passes (8 SQ slots each, MI355X_MICROARCH.md "rocprofv3 PMC slots"), PMC only with
# --kernel-trace.   usage: KIND=pdbbind|qm9|rna bash tools/pmc_issue.sh   ->  gpurun_out/issue_<kind>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KIND=${KIND:-pdbbind}
STEPS=${STEPS:-12}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES"
for p in 1 2; do
  eval "C=\$P$p"
  rm -rf /tmp/pmc_issue_$p
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_issue_$p -- python $R/tools/store_steps.py $KIND $STEPS serial > /tmp/pmc_issue_$p.log 2>&1
done
f1=$(find /tmp/pmc_issue_1 -name '*counter_collection.csv' | head -1)
f2=$(find /tmp/pmc_issue_2 -name '*counter_collection.csv' | head -1)
mkdir -p $R/gpurun_out
python - "$f1" "$f2" "$KIND" "$STEPS" <<'PY' > $R/gpurun_out/issue_$KIND.txt
import csv, sys, collections, re
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int); seen = set()
    for r in csv.DictReader(open(path)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'^void ', '', k)
        k = re.sub(r'\(.*', '', k)[:46]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Dispatch_Id'], k)
        if key not in seen:
            seen.add(key); cnt[k] += 1
    return acc, cnt
a1, c1 = load(sys.argv[1]); a2, c2 = load(sys.argv[2])
kind, steps = sys.argv[3], sys.argv[4]
print('# rocprofv3 --kernel-trace --pmc <pass> -- python tools/store_steps.py %s %s serial   (two passes; graph construction in line on the main stream)' % (kind, steps))
print('# pass 1: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM')
print('# pass 2: SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES')
print('# Of a wave\'s resident cycles (SQ_WAVE_CYCLES): parked = SQ_WAIT_ANY (s_waitcnt on memory / LDS, barriers), stalled = SQ_WAIT_INST_ANY')
print('# (wants to issue, the pipe is taken: MFMA read-after-write, a busy VALU / matrix pipe), issuing = SQ_ACTIVE_INST_ANY; the guide: the three')
print('# are disjoint and add up to ~the wave cycles.  An HBM-bound kernel is parked; an issue-bound one is stalled + issuing.')
print('# per wave: instructions issued by class (SQ_INSTS_* / SQ_WAVES).  mfma_busy/wave = SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVES (pipe cycles).')
print('%-46s %5s | %7s %7s %7s | %7s %7s %7s | %8s %7s %7s %7s %7s %9s' % ('kernel', 'calls', 'parked', 'stalled', 'issuing', 'valu', 'lds', 'vmem', 'VALU/wv', 'MFMA/wv', 'LDS/wv', 'RD/wv', 'WR/wv', 'mfma_busy'))
rows = []
for k, c in a1.items():
    wc = c.get('SQ_WAVE_CYCLES', 0)
    if wc <= 0 or k not in a2: continue
    d = a2[k]; w = d.get('SQ_WAVES', 0) or 1
    rows.append((wc, k, c1[k], c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                 c.get('SQ_ACTIVE_INST_VALU', 0) / wc, c.get('SQ_ACTIVE_INST_LDS', 0) / wc, c.get('SQ_ACTIVE_INST_VMEM', 0) / wc,
                 d.get('SQ_INSTS_VALU', 0) / w, d.get('SQ_INSTS_MFMA', 0) / w, d.get('SQ_INSTS_LDS', 0) / w, d.get('SQ_INSTS_VMEM_RD', 0) / w,
                 d.get('SQ_INSTS_VMEM_WR', 0) / w, d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w))
for r in sorted(rows, reverse=True)[:14]:
    print('%-46s %5d | %6.1f%% %6.1f%% %6.1f%% | %6.1f%% %6.1f%% %6.1f%% | %8.0f %7.0f %7.0f %7.0f %7.0f %9.0f' % ((r[1], r[2]) + tuple(100 * v for v in r[3:9]) + r[9:]))
PY
cat $R/gpurun_out/issue_$KIND.txt
