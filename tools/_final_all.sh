cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
{
python3 -c "from fasttrack_amd import orb; print('library', orb.version())"
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== tests/tools/soak_parity.py --trials 900 --seed 9931"; timeout 900 python3 tests/tools/soak_parity.py --trials 900 --seed 9931 2>&1 | tail -1
echo "== tests/tools/soak_search.py --trials 600 --seed 9932"; timeout 600 python3 tests/tools/soak_search.py --trials 600 --seed 9932 2>&1 | tail -2
echo "== tests/tools/soak_batch.py --trials 200 --seed 9933 --frames 32"; timeout 900 python3 tests/tools/soak_batch.py --trials 200 --seed 9933 --frames 32 2>&1 | tail -1
echo "== tests/tools/soak_batch.py --trials 200 --seed 9934 --frames 6"; timeout 600 python3 tests/tools/soak_batch.py --trials 200 --seed 9934 --frames 6 2>&1 | tail -1
} > gpurun_out/final/r06_soak_and_tests_final.txt 2>&1
timeout 2400 bash tools/final_measure.sh r06 > gpurun_out/final_r06.log 2>&1
tail -30 gpurun_out/final/r06_soak_and_tests_final.txt
