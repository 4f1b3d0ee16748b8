cd /root/repo; export TMPDIR=/tmp
D=/tmp/off_trace; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/traverse_replay.py profiles/r02_sampling_state.npz 200 > $D/out.txt 2>&1
tail -1 $D/out.txt
python tools/kernel_summary.py $D | grep -E "traverse_" | cut -c1-160
python tools/traverse_replay.py profiles/r02_sampling_state.npz 3 --check | tail -1
python -m pytest tests/test_k2_reference.py tests/test_gpu_grid.py -x -q 2>&1 | tail -2
