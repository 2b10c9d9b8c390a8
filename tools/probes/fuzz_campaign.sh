# Round-end fuzz campaign on seeds the suite does not use: device vs oracle (beam search in both LM behaviours, front end, audio
# ingest, random architectures x batch classes through the product's own kernel choice).
# bash tools/probes/fuzz_campaign.sh <tag> [seconds per fuzzer] [seed0]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-fuzz}; S=${2:-500}; S0=${3:-100000}; mkdir -p $O; cd $R
python tests/devtools/fuzz_encoder.py 100000 $S0 $S > $O/fuzz_encoder.log 2>&1
python tests/devtools/fuzz_frontend.py 100000 $S0 $S > $O/fuzz_frontend.log 2>&1
python tests/devtools/fuzz_beam.py 100000 $S0 $S > $O/fuzz_beam.log 2>&1
python tests/devtools/fuzz_audio.py 100000 $S0 $((S/3)) > $O/fuzz_audio.log 2>&1
for f in encoder frontend beam audio; do echo "== $f"; grep -i "mismatch\|error\|Traceback" $O/fuzz_$f.log | head -5 | cut -c1-400; tail -1 $O/fuzz_$f.log | cut -c1-400; done
