# usage: tools/r4_gen_force.sh N "L1,L2 L1,L2 ..." [d]: per-pass kernel times of complex N on forced tile lengths (g suffix = run-time plan)
N=$1; shift; PL=$1; shift
for pl in $PL; do
  echo "--- $N plan $pl"
  PFFFT_HIP_TILE_FORCE=$pl bash tools/kstats.sh gf python r4_gen_prof.py $N "$@" | grep "tile" | sed -e 's/(pf::vec2t.*calls/ calls/'
done
