cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p_ri
cat > /tmp/ri.py <<'PY'
import os, sys, time
R=os.environ['GRAFT_REPO_ROOT']
for p in (R, os.path.join(R, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth, store as S
dev=torch.device('cuda:0'); torch.manual_seed(0)
model = models.PAMNet(models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')).to(dev).eval()
graphs=[synth.rna_chain(2, i) for i in range(8)]
st=S.MoleculeStore(graphs, dev).prepare_for(model)
idx=[[(i+2*k)%8 for i in range(8)] for k in range(4)]
with torch.no_grad():
    for k in range(4): model(st.collate(idx[k]))
    torch.cuda.synchronize(); t0=time.perf_counter()
    for k in range(100): model(st.collate(idx[k%4]))
    torch.cuda.synchronize()
print('rna store forward (no grad, unpipelined): %.3f ms' % ((time.perf_counter()-t0)/100*1e3))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ri -- python /tmp/ri.py > /tmp/p_ri.log 2>&1
grep 'ms' /tmp/p_ri.log | tail -1
f=$(find /tmp/p_ri -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows[:32]:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])/104/1e3
    tot+=t
    print('%8.1f us/fwd  x%4.1f  %s' % (t, n/104, r['Name'][:90]))
print('sum of listed: %.1f us/fwd' % tot)
PY
