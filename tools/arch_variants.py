import os, sys, importlib.util, warnings
warnings.simplefilter("ignore")
here = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, here)
spec_ = importlib.util.spec_from_file_location("arch_shape_opt", os.path.join(here, "examples", "arch_shape_opt.py"))
mod = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(mod)
out = mod.run(verbose=False, p=2)
print("GF_WALK=%s h1 %.6f w1 %.6e newton rel res %.3e" % (os.environ.get("GF_WALK"), out["h1"], out["w1"], out["problem"].nm.newton_relative_residual if hasattr(out["problem"], "nm") else -1))
