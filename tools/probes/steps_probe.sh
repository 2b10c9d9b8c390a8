cd $GRAFT_REPO_ROOT
for k in 20 200 1000 20; do
python bench.py --steps $k --warmup 3 --no-cpu-baseline --no-other-gemm --no-side-configs 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('steps', j['steps'], 'ms', j['ms_per_step'], 'gemm', j['roofline']['ms_per_step'], j['roofline']['frac'], 'dw', j['depthwise']['ms_per_step'], 'fused', j['fused']['ms_per_step'], 'box', j['box']['measured_mfma_tflops'])"
done
