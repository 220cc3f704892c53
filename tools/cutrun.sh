python tools/landscape.py --genes 60000 --ot 0.5 --k 31 --q 20 --reps 3 2>/dev/null
python tools/landscape.py --genes 1000,60000 --ot 0.0,0.5 --reps 3 2>/dev/null
