import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from conv_bench import run
for nb in [6, 48, 126, 252, 378, 510, 636, 768, 900, 1024]:
    run(768, nb // 6 * 128, [7], 0, label=f"{nb} blocks")
