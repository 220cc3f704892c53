python tools/landscape.py --genes 1 --ot 0.0,0.5,1.0 --reps 4 2>/dev/null
python tools/landscape.py --genes 1000,60000 --ot 0.5 --reps 3 2>/dev/null
python tools/landscape.py --genes 1 --ot 0.5 --read-len 300 --pairs 5000000 --reps 3 2>/dev/null
