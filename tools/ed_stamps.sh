# builds the nine diagnostic libraries of tools/ed_stamps.py (here, before gpurun): bash tools/ed_stamps.sh
cd "$(dirname "$0")/../lane_slam_amd/csrc" || exit 1
for k in ${@:-1 2 3 4 5 6 7 8 9 10}; do
  rm -f _build_ed/k_edlines.o
  make -j8 EXTRA=-DLF_ED_STAMP=$k BUILD=_build_ed OUT=../liblanefront_ed$k.so 2>&1 | grep -E "error" 
done
ls -la ../liblanefront_ed*.so | wc -l
