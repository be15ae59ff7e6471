cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_out -- python $R/tools/narrow_bench2.py > /dev/null 2>&1
  f=$(find /tmp/pmc_out -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'nlinear_bwd' not in k: continue
    acc[r['Counter_Name']]['v'].append(float(r['Counter_Value']))
for c,v in acc.items():
    vals=v['v']; n=len(vals)//4
    # four variants in order (act=1,dx=1),(1,0),(0,1),(0,0), 23 launches each
    print(c, ['%.3g'%(sum(vals[i*n:(i+1)*n])/n) for i in range(4)])
PY
done
