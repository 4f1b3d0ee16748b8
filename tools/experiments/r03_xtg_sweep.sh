cd /root/repo
R="python tools/traverse_replay.py profiles/r02_sampling_state.npz 30"
for n in 4096 6500 8192 13000 20000 32000 50000 65536 100000 200000; do
  echo "== $n default";          $R --rays=$n 2>&1 | tail -1
  echo "== $n P16 L2";           NFA_SPLIT_L2=1 NFA_SPLIT_P=16 $R --rays=$n 2>&1 | tail -1
  echo "== $n P16 L2 XT(LDS)";   NFA_SPLIT_L2=1 NFA_SPLIT_L2_XT=1 NFA_SPLIT_P=16 $R --rays=$n 2>&1 | tail -1
  echo "== $n P8 L2";            NFA_SPLIT_L2=1 NFA_SPLIT_P=8 $R --rays=$n 2>&1 | tail -1
  echo "== $n P4 L2";            NFA_SPLIT_L2=1 NFA_SPLIT_P=4 $R --rays=$n 2>&1 | tail -1
  echo "== $n P1";               NFA_SPLIT_P=1 $R --rays=$n 2>&1 | tail -1
  echo "== $n P1 L2";            NFA_SPLIT_P=1 NFA_COUNT_L2=1 $R --rays=$n 2>&1 | tail -1
done
