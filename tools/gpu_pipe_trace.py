"""Two iterations of BASELINE config[1] through the pipelined pcut loop and nothing else: the command for
`rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/gpu_pipe_trace.py [N] [long_draws] [iterations]`; tools/pipe_trace_summary.py turns
the kernel trace into the overlap figures of profiles/r04_pipelined_pcuts_trace.txt."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import mcs, make_problem, hip_backend

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 2          # (more iterations: a soak run of the pipelined loop)
prob = make_problem(N, num_iterations=IT)
hb = hip_backend(prob)
r = mcs.driver.run(prob, hb, n_itrs=IT, long_draws=B, species_tallies="light")
print("steps", r.steps_helix + r.steps_retro, "pcuts", len(r.stats))
hb.destroy()
