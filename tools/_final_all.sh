cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
{
python3 -c "from fasttrack_amd import orb; print('library', orb.version())"
echo "== pytest -m gpu"; python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== tests/tools/soak_parity.py --trials 2000 --seed 9701"; python3 tests/tools/soak_parity.py --trials 2000 --seed 9701 2>&1 | tail -1
echo "== tests/tools/soak_search.py --trials 2500 --seed 9702"; python3 tests/tools/soak_search.py --trials 2500 --seed 9702 2>&1 | tail -2
echo "== tests/tools/soak_batch.py --trials 300 --seed 9703 --frames 32"; python3 tests/tools/soak_batch.py --trials 300 --seed 9703 --frames 32 2>&1 | tail -2
echo "== tests/tools/soak_batch.py --trials 300 --seed 9704 --frames 12"; python3 tests/tools/soak_batch.py --trials 300 --seed 9704 --frames 12 2>&1 | tail -2
} > gpurun_out/final/r06_soak_and_tests.txt 2>&1
bash tools/final_measure.sh r06 > gpurun_out/final_r06.log 2>&1
tail -30 gpurun_out/final/r06_soak_and_tests.txt
