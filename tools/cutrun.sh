python tools/landscape.py --genes 1000,60000 --read-len 250 --ot 0.5 --pairs 5000000 --reps 3 2>/dev/null
