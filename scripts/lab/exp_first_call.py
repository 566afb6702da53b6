"""Where does the FIRST solve() of a process spend its time (module load, allocator warm-up)?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t00 = time.perf_counter()
import numpy as np
from padne_amd import mesh, problem, solver, structured
size = 0.6
W, H = 420.0, 420.0
top = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, W, H)), name="F.Cu", conductance=2082.5)
bot = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, W, H)), name="B.Cu", conductance=2082.5)
P = mesh.Point
rng = np.random.default_rng(0)
nets = []
for k in range(200):
    x, y = rng.uniform(5, W - 5), rng.uniform(5, H - 5)
    a, b = problem.NodeID(), problem.NodeID()
    nets.append(problem.Network(connections=[problem.Connection(top, P(x, y), a), problem.Connection(bot, P(x, y), b)],
                                elements=[problem.Resistor(a, b, 1e-3)]))
s_p, s_n, l_a, l_b = problem.NodeID(), problem.NodeID(), problem.NodeID(), problem.NodeID()
nets.append(problem.Network(connections=[problem.Connection(top, P(10, 10), s_p), problem.Connection(bot, P(10, 10), s_n)],
                            elements=[problem.VoltageSource(s_p, s_n, 1.0)]))
nets.append(problem.Network(connections=[problem.Connection(top, P(W - 10, H - 10), l_a), problem.Connection(bot, P(W - 10, H - 10), l_b)],
                            elements=[problem.Resistor(l_a, l_b, 0.05)]))
prob = problem.Problem(layers=[top, bot], networks=nets)
mesher = structured.StructuredMesher(mesh.Mesher.Config(maximum_size=size), jitter=0.2)
print(f"imports + problem: {time.perf_counter() - t00:.3f} s", flush=True)
t0 = time.perf_counter(); ctx = solver.get_context(); print(f"context: {time.perf_counter() - t0:.3f} s", flush=True)
pr = cProfile.Profile(); pr.enable(); sol = solver.solve(prob, mesher=mesher); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
