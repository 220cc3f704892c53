for n in 3 4; do echo "CUT $n"; SHK_LIB_PATH=$PWD/tools/variants/cut$n.so python tools/landscape.py --genes 60000 --ot 1.0 --reps 3 2>/dev/null; done
echo STATS; SHK_LIB_PATH=$PWD/tools/variants/anchstats.so python tools/landscape.py --genes 60000 --ot 1.0,0.5,0.0 --reps 1 2>/dev/null
