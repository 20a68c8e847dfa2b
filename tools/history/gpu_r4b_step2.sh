#!/bin/bash
# round 4 (second session), second GPU call: A/B of builds (step tail, sum of weights, non-temporal list loads / stores),
# of the engines' merge threshold, and cache / texture-addresser counters of the list passes
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOTDIR
mkdir -p gpurun_out
timeout 1500 python tools/gpu_abx_libs.py 2 libcvo_hip.so libcvo_hip_base0.so libcvo_hip_suma.so libcvo_hip_nt.so libcvo_hip_ntst.so -- "10000 6 64" "10000 3 256" 2>&1 | tee gpurun_out/r4b_ab_libs.txt
timeout 1500 python tools/gpu_abx.py 2 "CVO_HIP_MERGE_MAX=2" "CVO_HIP_MERGE_MAX=4" "CVO_HIP_MERGE_MAX=8" "CVO_HIP_MERGE_MAX=16" "CVO_HIP_MERGE_MAX=32" -- "10000 6 64" "10000 3 256" 2>&1 | tee gpurun_out/r4b_ab_merge.txt
OUT=$ROOTDIR/gpurun_out/r4b_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOTDIR/tools/gpu_batch.py 10000 2 64"
i=0
for SET in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  DISTINCT=1 CVO_HIP_GRAPH=1 timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
  tail -2 $OUT/p$i.log
done
cd $ROOTDIR
python - <<PY
import csv,collections,glob,json
out={}
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    per=collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> dispatch -> counter -> value
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('cvo_dev::','').replace('void ','')
        per[k][r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
    for k,d in per.items():
        if 'rocclr' in k: continue
        rows=list(d.values())
        key='GRBM_GUI_ACTIVE' if 'GRBM_GUI_ACTIVE' in rows[0] else 'SQ_BUSY_CYCLES'
        rows.sort(key=lambda x:-x.get(key,0))
        top=rows[:max(1,len(rows)//20)]
        for name,sel in (('all',rows),('top5pct',top)):
            tot=collections.defaultdict(float)
            for x in sel:
                for c,v in x.items(): tot[c]+=v
            out.setdefault(k,{}).setdefault(name,{}).update({c:v for c,v in tot.items()}); out[k][name]['n_'+f.split('/')[-2]]=len(sel)
json.dump(out,open("$ROOTDIR/gpurun_out/r4b_pmc_summary.json","w"),indent=1)
for k,v in out.items():
    for name,t in v.items():
        hit=t.get('TCC_HIT_sum'); miss=t.get('TCC_MISS_sum')
        s="%-22s %-8s" % (k[:22],name)
        if hit is not None and hit+miss>0: s+=" L2hit %.3f req %.3g" % (hit/(hit+miss), t.get('TCC_REQ_sum',0))
        if 'TA_TA_BUSY_sum' in t and t.get('GRBM_GUI_ACTIVE'): s+=" TA_busy/(256*GUI) %.3f TCPstall/(256 GUI) %.3f tcp_acc %.3g tcc_rd %.3g" % (t['TA_TA_BUSY_sum']/256/t['GRBM_GUI_ACTIVE'], t.get('TCP_PENDING_STALL_CYCLES_sum',0)/256/t['GRBM_GUI_ACTIVE'], t.get('TCP_TOTAL_CACHE_ACCESSES_sum',0), t.get('TCP_TCC_READ_REQ_sum',0))
        if 'SQ_WAVE_CYCLES' in t: s+=" valu/wavecyc %.3f waitinst %.3f waitany %.3f mfma_busy/busy %.3f" % (t['SQ_ACTIVE_INST_VALU']/t['SQ_WAVE_CYCLES'], t['SQ_WAIT_INST_ANY']/t['SQ_WAVE_CYCLES'], t['SQ_WAIT_ANY']/t['SQ_WAVE_CYCLES'], t.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(1,t['SQ_BUSY_CYCLES']))
        print(s)
PY
