for rep in 1 2; do
for lib in build/libmppi_head.so mppi_numba_amd/libmppi_hip.so; do
  for cfg in "2 4096" "3 2048"; do set -- $cfg
    MPPI_HIP_LIB=$PWD/$lib timeout 200 python tools/p2p_ranks.py --ranks $1 --n $2 --t 100 --iterations 20 --time 400 2>&1 | grep -E "P2P_" | sed "s/.*max|du|/max|du|/" | sed "s|^|$lib ranks=$1 n=$2: |"
  done
done
done
