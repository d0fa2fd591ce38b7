import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from conv_bench import run
run(768, 131072, [7], 0, label="8 rounds", reps=3)
run(768, 131072 * 2, [7], 0, label="16 rounds", reps=3)
