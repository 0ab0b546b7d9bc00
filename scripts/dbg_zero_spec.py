"""Upper bound for specialising fm_voice on its zero-valued default parameters (feedback, route, filter-envelope
amount): the same graph with those paths removed by hand, timed against the built-in on the driver's workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oscen_amd

N, BLOCK, W, K = 65536, 256, 5, 20
dsl = oscen_amd.Graph(builtin="fm_voice").to_dsl()
simp = dsl
for line in ("    op3_feedback -> op3_osc.feedback;\n", "    op2_feedback -> op2_osc.feedback;\n",
             "    env_filter.output -> filter_env_gain.input;\n", "    filter_env_amount -> filter_env_gain.gain;\n",
             "    filter_env_gain.output -> cutoff_mod.input;\n", "    filter_cutoff -> cutoff_mod.value;\n",
             "    cutoff_mod.output -> filter.cutoff;\n", "    op3_osc.output -> op3_route.input;\n", "    route -> op3_route.mix;\n",
             "    op3_route.output_a -> op2_osc.phase_mod;\n", "    op2_osc.output -> op1_mod_mixer.input_a;\n",
             "    op3_route.output_b -> op1_mod_mixer.input_b;\n", "    op1_mod_mixer.output -> op1_osc.phase_mod;\n"):
    assert line in simp, line
    simp = simp.replace(line, "")
simp = simp.replace("connections {\n", "connections {\n    filter_cutoff -> filter.cutoff;\n    op3_osc.output -> op2_osc.phase_mod;\n"
                    "    op2_osc.output -> op1_osc.phase_mod;\n" + ("    env_filter.output * 0.0 -> op1_osc.feedback;\n" if "keepenv" in sys.argv else ""))
plans = oscen_amd.note_plans(N, span=(W + K) * BLOCK, fold="slice")


def run(name, g):
    e = oscen_amd.Engine(g, N, sample_rate=48000.0)
    oscen_amd.schedule_note_plans(e, plans, total_frames=(W + K) * BLOCK)
    e.set_bus_batching(0)
    for _ in range(W):
        e.process_block_async(BLOCK)
    e.flush(); e.synchronize()
    e.enable_kernel_timing(True)
    for _ in range(K):
        e.process_block_async(BLOCK)
    e.flush(); e.synchronize()
    ms, n = e.kernel_time_ms()
    print("%-28s %s depth %d  kernel ms/block %.5f" % (name, e.kernel_variant, e.pipeline_depth, ms * n / K), flush=True)


run("built-in fm_voice", "fm_voice")
run("same from DSL (JIT)", oscen_amd.Graph(dsl=dsl, per_voice=("frequency",)))
run("zero paths removed (JIT)", oscen_amd.Graph(dsl=simp, per_voice=("frequency",)))
