"""cProfile of the whole Python seam padne_amd.solver.solve() on a Problem-level input with ~1 M vertices
(two layers, via-like resistors, a voltage source and a load) -- where does host time go outside the device?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from padne_amd import mesh, problem, solver, structured

size = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6
W, H = 420.0, 420.0
top = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, W, H)), name="F.Cu", conductance=2082.5)
bot = problem.Layer(shape=structured.Shapes.of(structured.Rect(0, 0, W, H)), name="B.Cu", conductance=2082.5)
P = mesh.Point
rng = np.random.default_rng(0)
nets = []
conns = lambda layer, x, y: problem.Connection(layer=layer, point=P(x, y))
vias = []
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
t0 = time.perf_counter(); sol = solver.solve(prob, mesher=mesher); t1 = time.perf_counter()
nv = sum(len(ls.meshes[0].vertices) if hasattr(ls.meshes[0], "vertices") else 0 for ls in sol.layer_solutions)
print(f"first solve() {t1 - t0:.3f} s", flush=True)
for _ in range(2):
    t0 = time.perf_counter(); sol = solver.solve(prob, mesher=mesher); print(f"solve() again {time.perf_counter() - t0:.3f} s", flush=True)
pr = cProfile.Profile(); pr.enable(); sol = solver.solve(prob, mesher=mesher); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
