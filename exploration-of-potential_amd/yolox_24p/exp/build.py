"""``get_exp(exp_file)``: import the file by path and instantiate its ``Exp`` (reference exp/build.py:6-30)."""
import importlib
import os
import sys


def get_exp_by_file(exp_file):
    try:
        sys.path.append(os.path.dirname(exp_file))
        module = importlib.import_module(os.path.basename(exp_file).split(".")[0])
        exp = module.Exp()
    except Exception:
        raise ImportError("{} doesn't contains class named 'Exp'".format(exp_file))
    return exp


def get_exp(exp_file):
    assert exp_file is not None, "plz provide exp file."
    return get_exp_by_file(exp_file)
