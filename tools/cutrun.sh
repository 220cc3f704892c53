python tools/landscape.py --genes 1,1000,60000 --ot 0.0,0.5,1.0 --reps 3 2>/dev/null
