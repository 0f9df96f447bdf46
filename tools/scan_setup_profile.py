"""Where the per-point setup of a grid scan goes (scan._SensPoint / _TexturePoint construction)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.getcwd())
from golemflavor_amd import scan
from golemflavor_amd.enums import Texture
pts = scan.sens_grid()
def c5():
    jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]
    return jobs
jobs = c5(); [j.close() for j in jobs]
t0 = time.perf_counter(); jobs = c5(); print("C5: %d points in %.1f ms" % (len(pts), 1e3 * (time.perf_counter() - t0))); [j.close() for j in jobs]
pr = cProfile.Profile(); pr.enable(); jobs = c5(); pr.disable(); [j.close() for j in jobs]
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
pts4 = scan.texture_grid(6)
def c4():
    return [scan._TexturePoint(p, g, dimension=6, texture=Texture.OET, nwalkers=2048, device=0) for g, p in enumerate(pts4)]
jobs = c4(); [j.close() for j in jobs]
t0 = time.perf_counter(); jobs = c4(); print("C4: %d points in %.1f ms" % (len(pts4), 1e3 * (time.perf_counter() - t0))); [j.close() for j in jobs]
pr = cProfile.Profile(); pr.enable(); jobs = c4(); pr.disable(); [j.close() for j in jobs]
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
