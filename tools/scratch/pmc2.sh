cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_r2b
mkdir -p $OUT
OP=fwd
for shape in "256 256 3 1 14" "64 64 3 1 56"; do
 tag=$(echo $shape | tr ' ' '_')_$OP
 p=0
 for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TA_BUSY_avr TCC_BUSY_avr GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
  p=$((p+1))
  ICAMD_CONV3X3_HALO=3 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/${tag}_p$p -- python3 $R/tools/one_layer.py $shape 4 $OP > $OUT/${tag}_p$p.log 2>&1
 done
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_r2b'
for d in sorted(glob.glob(out+'/*_p*')):
    if not os.path.isdir(d): continue
    fs=glob.glob(d+'/*/*counter_collection.csv')
    if not fs: print(d,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name']
        if 'conv' not in k: continue
        k=k.split('(')[0][-40:]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    for k,v in acc.items():
        n=len(cnt[k])
        print(os.path.basename(d), k, 'n=%d'%n, ' '.join('%s=%.4g'%(c,x/n) for c,x in sorted(v.items())))
PY
