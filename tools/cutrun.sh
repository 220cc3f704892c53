for a in 0 64 32 16; do echo "ABL $a"; SHK_ABLATE=$a SHK_LIB_PATH=$PWD/tools/variants/abl.so python tools/landscape.py --genes 60000 --ot 1.0 --reps 3 2>/dev/null; done
