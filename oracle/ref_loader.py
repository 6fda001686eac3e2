"""Load the reference planner classes from /root/reference WITHOUT running their
module-level driver code (test infrastructure; build container only).

The reference scripts are cell-style: digit-leading file names, no __main__
guard, animation + plt.show() at import.  Following SURVEY.md section 8(c), the
file is read as text, parsed with `ast`, and only imports, function/class
definitions and the plain module-level assignments the definitions need (Sobol
globals rrt_04:41-51, _PATH_TYPE_MAP rrt_05:1797, show_animation) are executed
in a fresh module object.  Nothing from the reference is copied into this repo;
/root/reference does not exist on the GPU box, so only golden-generation
scripts use this.
"""
import ast
import os
import sys
import types

REF_DIR = "/root/reference/src_path_planning"

FILES = {
    "rrt_01": "10_path_planning_01_rrt_01_simple.py",
    "rrt_02": "10_path_planning_01_rrt_02_sobol_sampler.py",
    "rrt_03": "10_path_planning_01_rrt_03_dubins_path.py",
    "rrt_04": "10_path_planning_01_rrt_04_rrt_star.py",
    "rrt_05": "10_path_planning_01_rrt_05_rrt_star_dubins_path.py",
    "rrt_06": "10_path_planning_01_rrt_06_rrt_star_reeds_shepp_path.py",
    "rrt_07": "10_path_planning_01_rrt_07_informed_rrt_star.py",
    "rrt_08": "10_path_planning_01_rrt_08_batch_informed_rrt_star.py",
}

KEEP_ASSIGN = {"atmost", "dim_max", "dim_num_save", "initialized", "lastq", "log_max", "maxcol", "poly",
               "recipd", "seed_save", "v", "_PATH_TYPE_MAP", "show_animation"}


def load(short):
    os.environ.setdefault("MPLBACKEND", "Agg")
    path = os.path.join(REF_DIR, FILES[short])
    src = open(path).read()
    tree = ast.parse(src, filename=path)
    body = []
    deferred = []
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom, ast.FunctionDef, ast.ClassDef)):
            body.append(node)
        elif isinstance(node, ast.Assign):
            names = [t.id for t in node.targets if isinstance(t, ast.Name)]
            if names and all(n in KEEP_ASSIGN for n in names):
                # rrt_03 builds _PATH_TYPE_MAP (:1030) BEFORE the word functions it names are defined (:1138-1218),
                # so the script as shipped stops with NameError; the table is evaluated after the definitions here
                # (as rrt_05:1797 places it), nothing else is reordered
                (deferred if "_PATH_TYPE_MAP" in names else body).append(node)
    body.extend(deferred)
    mod = types.ModuleType("ref_" + short)
    mod.__file__ = path
    code = compile(ast.Module(body=body, type_ignores=[]), path, "exec")
    exec(code, mod.__dict__)
    if "show_animation" not in mod.__dict__:
        mod.show_animation = False
    mod.show_animation = False
    return mod


def reset_sobol(mod):
    """Clear the Sobol module-global state between runs (rrt_04:41-51, :299-309)."""
    if hasattr(mod, "initialized"):
        mod.initialized = None
    if hasattr(mod, "seed_save"):
        mod.seed_save = None
