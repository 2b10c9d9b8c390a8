# Final-state campaign: every fuzzer on seeds none of the earlier calls used.  bash tools/probes/fuzz_campaign2.sh <tag> [seconds per fuzzer] [seed0]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-fuzz2}; S=${2:-150}; S0=${3:-300000}; mkdir -p $O; cd $R
python tests/devtools/fuzz_frontend.py 100000 $S0 $S > $O/fuzz_frontend.log 2>&1
python tests/devtools/fuzz_encoder.py 100000 $S0 $S > $O/fuzz_encoder.log 2>&1
python tests/devtools/fuzz_beam.py 100000 $S0 $S > $O/fuzz_beam.log 2>&1
python tests/devtools/fuzz_audio.py 100000 $S0 $((S/3)) > $O/fuzz_audio.log 2>&1
python tests/devtools/fuzz_rows.py 100000 $S0 $S > $O/fuzz_rows.log 2>&1
python tests/devtools/fuzz_dag.py 100000 $S0 $S > $O/fuzz_dag.log 2>&1
python tests/devtools/fuzz_dag.py 100000 $S0 $((S/2)) beam > $O/fuzz_dag_beam.log 2>&1
python tests/devtools/fuzz_long.py 100000 $S0 $S > $O/fuzz_long.log 2>&1
python tests/devtools/stress_serving.py $((S/3)) 24 7 > $O/stress_serving.log 2>&1
python tests/devtools/stress_threads.py $((S/3)) 4 7 > $O/stress_threads.log 2>&1
for f in fuzz_frontend fuzz_encoder fuzz_beam fuzz_audio fuzz_rows fuzz_dag fuzz_dag_beam fuzz_long stress_serving stress_threads; do echo "== $f"; grep -i "mismatch\|Traceback" $O/$f.log | head -3 | cut -c1-300; grep -v "amdgpu\|Warn\|warn" $O/$f.log | tail -1 | cut -c1-400; done
