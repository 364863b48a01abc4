"""Flags of the kept entry point that need no GPU (reference parser: yolox_24p/train_24p.py:180-201)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Y24 = os.path.join(ROOT, "exploration-of-potential_amd", "yolox_24p")


def test_cpu_device_is_refused(tmp_path):
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(Y24, "train_24p.py"), "-f", os.path.join(Y24, "load_train", "yolox_24p_train.py"),
                        "-b", "1", "--device", "cpu", "--steps", "1", "--output-dir", str(tmp_path)], cwd=Y24, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU fallback" in p.stdout, p.stdout[-2000:]


def test_parser_keeps_the_reference_flags():
    sys.path.insert(0, Y24)
    try:
        import importlib
        mod = importlib.import_module("train_24p")
        a = mod.make_parser().parse_args(["-f", "x.py", "-b", "20", "-l", "0.01", "-s", "1", "-d", "2", "--resume", "-c", "w.pth", "-e", "3",
                                          "--num_machines", "1"])
        assert (a.exp_file, a.batch_size, a.learn_rate, a.start_device, a.devices, a.resume, a.ckpt, a.start_epoch) == \
            ("x.py", 20, 0.01, 1, 2, True, "w.pth", 3)
        d = mod.make_parser().parse_args([])
        assert (d.batch_size, d.learn_rate, d.device, d.synthetic, d.steps) == (4, 0.001, "cuda", False, 0)
    finally:
        sys.path.remove(Y24)


def test_backbone_swap_exp_files():
    """BASELINE config 4 is reachable from the kept entry points: Exp.backbone_type -> YOLOPAFPN(backbone_type)."""
    sys.path.insert(0, Y24)
    try:
        from exp import get_exp
        for name, kind in (("yolox_24p_l_resnet_train.py", "ResNet"), ("yolox_24p_l_densenet_train.py", "DenseNet"), ("yolox_24p_l_train.py", "CSPDarknet")):
            exp = get_exp(os.path.join(Y24, "load_train", name))
            model = exp.get_model()
            assert type(model.backbone.backbone).__name__ == kind
            assert abs(float(model.head.cls_preds[0].bias[0]) + 4.59512) < 1e-4
    finally:
        sys.path.remove(Y24)
