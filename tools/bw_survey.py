"""Achieved HBM bandwidth of the streaming kernels at their default tilings on the step's main shapes (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1] + ["0"]
import micro_dw  # noqa: F401  prints the dw / gate table at default strip heights
import micro_pw
for sh in [(8, 190, 36, 60000), (8, 36, 95, 60000), (8, 36, 36, 60000), (8, 108, 36, 60000), (8, 382, 72, 15000), (8, 72, 191, 15000),
           (8, 766, 144, 3750), (8, 144, 383, 3750), (8, 36, 36, 240000), (8, 36, 72, 240000)]:
    us, tbs, tf = micro_pw.run(*sh, flags=0)
    print(f"pw fwd B,M,K,HW={sh}: {us:7.1f} us {tbs:.2f} TB/s {tf:.1f} TF", flush=True)
