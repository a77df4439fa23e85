cd /root/repo; mkdir -p gpurun_out
export MIOPEN_USER_DB_PATH=/tmp/uh_miopen_udb; rm -rf $MIOPEN_USER_DB_PATH; mkdir -p $MIOPEN_USER_DB_PATH
B="python bench.py --quality 0 --cpu_baseline 0 --traffic 0 --north_star 0 --config4 0 --steps 60"
t0=$(date +%s)
# baseline with an empty user db (find mode as shipped)
timeout 300 $B 2>gpurun_out/r05i_tune_base.err | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(json.dumps({'mode':'find as shipped (empty user db)','pairs_s':d['value'],'ms_per_step':d['ms_per_step'],'warmup_s':d['config']['warmup_seconds']}))" | tee gpurun_out/r05i_miopen_tuning_probe.jsonl
echo "base done $(( $(date +%s) - t0 )) s"
rm -rf $MIOPEN_USER_DB_PATH; mkdir -p $MIOPEN_USER_DB_PATH
# exhaustive tuning of every tunable solver MIOpen's find tries (MIOPEN_FIND_ENFORCE=SEARCH_DB_UPDATE), bounded by a timeout
t1=$(date +%s)
MIOPEN_FIND_ENFORCE=4 timeout 540 $B 2>gpurun_out/r05i_tune_search.err | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(json.dumps({'mode':'MIOPEN_FIND_ENFORCE=4 (search + db update) in this run','pairs_s':d['value'],'ms_per_step':d['ms_per_step'],'warmup_s':d['config']['warmup_seconds']}))" | tee -a gpurun_out/r05i_miopen_tuning_probe.jsonl
echo "search rc $? $(( $(date +%s) - t1 )) s"; ls -la $MIOPEN_USER_DB_PATH | head; du -sh $MIOPEN_USER_DB_PATH
# a second run that only READS the tuned user db
timeout 300 $B 2>gpurun_out/r05i_tune_reuse.err | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(json.dumps({'mode':'find with the tuned user perf-db','pairs_s':d['value'],'ms_per_step':d['ms_per_step'],'warmup_s':d['config']['warmup_seconds']}))" | tee -a gpurun_out/r05i_miopen_tuning_probe.jsonl
mkdir -p gpurun_out/r05i_udb; cp -r $MIOPEN_USER_DB_PATH/* gpurun_out/r05i_udb/ 2>/dev/null; ls gpurun_out/r05i_udb | head
tail -3 gpurun_out/r05i_tune_search.err | cut -c1-300
