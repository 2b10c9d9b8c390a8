# Round-end fuzz campaign on seeds the suite does not use: device vs oracle (beam search in both LM behaviours, front end, audio ingest).
# bash tools/probes/fuzz_campaign.sh <tag> [seconds per fuzzer]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-fuzz}; S=${2:-500}; mkdir -p $O; cd $R
python tests/devtools/fuzz_beam.py 100000 100000 $S > $O/fuzz_beam.log 2>&1
python tests/devtools/fuzz_frontend.py 100000 100000 $S > $O/fuzz_frontend.log 2>&1
python tests/devtools/fuzz_audio.py 100000 100000 $((S/2)) > $O/fuzz_audio.log 2>&1
for f in beam frontend audio; do echo "== $f"; grep -c . $O/fuzz_$f.log; grep -i "mismatch\|error\|Traceback\|FAIL" $O/fuzz_$f.log | head -5; tail -2 $O/fuzz_$f.log; done
