import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from conv_bench import run
run(768, 131072, [7], 0, label=sys.argv[1], reps=4)
run(192, 60000 * 4, [7, 7, 7], 1, label=sys.argv[1] + " 192x128", reps=4)
