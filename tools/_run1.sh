set -e
for o in 1 0; do
TORCHREC_AMD_MAIN_PRIORITY=1 TORCHREC_AMD_WGRAD_OVERLAP=$o timeout -k 10 300 python bench.py > gpurun_out/n1_p.json 2> gpurun_out/n1.err
echo hi-prio main, overlap $o: $(tail -1 gpurun_out/n1_p.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"])')
done
