for v in default tabw4; do echo "VARIANT $v"; if [ $v = default ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
python tools/landscape.py --genes 1000,60000 --ot 0.0,0.5,1.0 --reps 3 2>/dev/null
python tools/landscape.py --genes 60000 --ot 0.5 --k 31 --q 20 --reps 3 2>/dev/null
done
