"""exploration-of-potential_amd: MI355X-native hot path for IN2-ViAUn/Exploration-of-Potential (YOLOX-24p).

The directory name is not an importable identifier; add it to sys.path and import ``ep24`` (kernels +
host logic) or run the reference-compatible entry points under ``yolox_24p/`` from inside that directory.
"""
