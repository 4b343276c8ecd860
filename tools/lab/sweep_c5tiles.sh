R=$GRAFT_REPO_ROOT
for t in 0 256 512 1024; do
 if [ $t = 0 ]; then unset PP_TARGET_TILES; else export PP_TARGET_TILES=$t; fi
 for w in 0 8 16; do
 if [ $w = 0 ]; then unset PP_TILE_WAVES; else export PP_TILE_WAVES=$w; fi
 echo "== tiles=$t waves=$w"; python3 $R/tools/bench_vox.py --iters 100 --batch 4 --n 200000 --half 100 --P 30000 2>&1 | grep "kernels\|^batch" | cut -c1-110
 done
done
