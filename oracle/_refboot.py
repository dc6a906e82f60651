"""Bootstrap for importing the read-only reference (/root/reference) in THIS container only.

Test infrastructure. Used by oracle/pin_against_reference.py to (a) check the CPU restatement in
oracle/mtdgan_oracle.py against the real reference code and (b) generate tests/golden/*.npz.
Nothing here travels to the GPU box in a usable form: /root/reference does not exist there.
Recipe: SURVEY.md §8c (stub the absent third-party imports, never write bytecode into the reference).
"""
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def boot():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    _stub("cvxpy")
    _stub("pydicom")
    mo = _stub("monai")
    mo.inferers = _stub("monai.inferers", sliding_window_inference=lambda *a, **k: None)
    _stub("module.piq", FID=object)
    _stub("module.piq.feature_extractors", InceptionV3=object)
    import matplotlib  # noqa: F401  (engine imports pyplot)
    matplotlib.use("Agg")
