"""CPU tests of the host side: the C-ABI library builds, loads and exports every declared symbol; the
module surface keeps the reference's state_dict contract; the product path refuses to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch

import mtdgan_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as ge
    ge.build()
    from mtd_gan_amd import _lib
    return ctypes.CDLL(_lib.LIB_PATH)


def test_library_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "mtdgan_hip.h")).read()
    names = set(re.findall(r"\b(mtd_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 19
    missing = [n for n in sorted(names) if not hasattr(built_lib, n)]
    assert not missing, missing
    built_lib.mtd_version.restype = ctypes.c_char_p
    assert b"gfx950" in built_lib.mtd_version()


def test_argument_errors_without_gpu(built_lib):
    from mtd_gan_amd._lib import ConvArgs
    a = ConvArgs()
    built_lib.mtd_conv_igemm.restype = ctypes.c_int
    assert built_lib.mtd_conv_igemm(ctypes.byref(a), None) == -1          # MTD_EINVAL: null pointers
    assert built_lib.mtd_conv_igemm(None, None) == -1


def test_generator_state_dict_contract_and_cpu_refusal():
    from mtd_gan_amd.arch.Ours.networks import FFT_ConvBlock, ResFFT_Generator
    G = ResFFT_Generator(1, 32, 10, 3, 1)
    sd = G.state_dict()
    shapes = orc.g_param_shapes()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    assert sum(p.numel() for p in G.parameters()) == 467137
    # reference init quirk: Conv2d re-initialised N(0, 0.01), ConvTranspose2d left at default
    assert G.encoder[3].weight.std().item() < 0.02 and G.decoder[3].weight.std().item() > 0.02
    assert len(list(G.shared_parameters())) == 44 and G.task_specific_parameters() is None
    with pytest.raises(RuntimeError):
        G(torch.zeros(1, 1, 64, 64))
    with pytest.raises(RuntimeError):
        FFT_ConvBlock(32)(torch.zeros(1, 32, 64, 64))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mtd-gan_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "mtdgan_oracle" not in src and "import oracle" not in src, os.path.join(dp, f)


def test_fused_adamw_loads_reference_optimizer_checkpoint():
    """train.py:146-159 checkpoints torch.optim.AdamW state; FusedAdamW must resume from it (host-side conversion only)."""
    import torch
    from mtd_gan_amd.optimizers import FusedAdamW
    w = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    ref = torch.optim.AdamW(w, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    for p in w:
        p.grad = torch.randn_like(p)
    ref.step()
    ref.step()
    sd = ref.state_dict()
    mine = FusedAdamW(w, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    mine.load_state_dict(sd)
    for p in w:
        st = mine.state[p]
        assert st["step"] == 2 and isinstance(st["step"], int)
        assert torch.equal(st["exp_avg"], ref.state[p]["exp_avg"]) and torch.equal(st["exp_avg_sq"], ref.state[p]["exp_avg_sq"])
    assert mine.param_groups[0]["lr"] == 1e-4 and mine.param_groups[0]["weight_decay"] == 5e-4
    # and back: a checkpoint written by FusedAdamW resumes under torch.optim.AdamW (reference train.py:146-159)
    back = torch.optim.AdamW(w, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    back.load_state_dict(mine.state_dict())
    back.step()
    assert int(back.state[w[0]]["step"]) == 3
    assert isinstance(mine.state[w[0]]["step"], int)          # exporting did not disturb the live state


def test_derived_view_cache_keeps_one_generation():
    """kernels._remember: a new version of the same weight view replaces the cached one (torch optimizers bump the version
    counter every step; FusedAdamW drops entries itself through weights_changed)."""
    import importlib
    K = importlib.import_module("mtd_gan_amd.kernels")
    cache = {}
    for version in range(5):
        K._remember(cache, ("ptr", 32, 32, 288, 9), ("ptr", version, 0, 32, 32, 288, 9), ("packed", version))
        K._remember(cache, ("other", 32, 32, 9, 288), ("other", version, 0, 32, 32, 9, 288), ("packed2", version))
    assert len(cache) == 2 and cache[("ptr", 4, 0, 32, 32, 288, 9)] == ("packed", 4)


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The argument structs the Python host fills (mtd_gan_amd/_lib.py) have the size and field offsets the C header
    declares (gcc, no GPU): conv / weight-gradient arguments and the descriptor tables of the deferred reductions."""
    import subprocess
    from mtd_gan_amd import _lib as L
    checks = [("mtd_conv_args", L.ConvArgs, ["out", "mask", "ws", "scale2", "out2", "out2_ld"], {"inp": "in"}),
              ("mtd_wgrad_args", L.WgradArgs, ["p", "q", "dw", "db", "accumulate", "ws"], {}),
              ("mtd_wgrad_reduce_desc", L.WgradReduceDesc, ["a", "T", "nslab", "slab_stride", "first_block"], {}),
              ("mtd_mix_reduce_desc", L.MixReduceDesc, ["ws", "dw2", "db2", "nslab", "accumulate"], {}),
              ("mtd_pack_desc", L.PackDesc, ["src", "dst", "N", "T", "sn", "sc"], {}),
              ("mtd_prof_record", L.ProfRecord, ["kernel", "taps", "M", "flops", "ms", "bytes"], {}),
              ("mtd_sum_desc", L.SumDesc, ["a", "b", "na", "nb"], {}),
              ("mtd_zero_desc", L.ZeroDesc, ["p", "n"], {})]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/mtdgan_hip.h"', "int main(void) {"]
    for cname, _cls, fields, _ren in checks:
        lines.append(f'  printf("%zu", sizeof({cname}));')
        for f in fields:
            lines.append(f'  printf(" %zu", offsetof({cname}, {f}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for (cname, cls, fields, ren), line in zip(checks, out):
        vals = [int(v) for v in line.split()]
        inv = {v: k for k, v in ren.items()}
        assert vals[0] == ctypes.sizeof(cls), cname
        for f, off in zip(fields, vals[1:]):
            assert getattr(cls, inv.get(f, f)).offset == off, (cname, f)


def test_gradient_buffer_name_rules_cover_the_discriminator():
    """train_step's first-pass-overwrite and early-shipping rules select layers BY NAME: the trunk's spectral-norm weights
    (overwritten by the first pass into a task vector), the decoders' (overwritten in the task-specific bucket), and the
    contiguous tail of a task vector that trunk levels >= FLUSH_LEVEL + the bottleneck form.  Here against the module's own
    parameter partition: every shared weight is covered, nothing else is, and the tail is what the rule says."""
    from mtd_gan_amd import train_step as TS
    from mtd_gan_amd import discriminator_path as DP
    from mtd_gan_amd.arch.Ours.networks import Multi_Task_Discriminator_Skip
    D = Multi_Task_Discriminator_Skip(1, 64)
    by_id = {id(p): n for n, p in D.named_parameters()}
    shared = [by_id[id(p)] for p in D.shared_parameters()]
    tspec = [by_id[id(p)] for p in D.task_specific_parameters()]
    assert "c_fc.weight_orig" not in shared + tspec                       # the reference's omission (frozen), kept
    sh_w = [n for n in shared if n.endswith(".weight_orig")]
    assert len(sh_w) == 20 and all(TS._TRUNK_SN_WEIGHT.match(n) for n in sh_w)
    assert not any(TS._TRUNK_SN_WEIGHT.match(n) for n in shared if not n.endswith(".weight_orig"))
    assert not any(TS._TRUNK_SN_WEIGHT.match(n) for n in tspec)
    dec = {pre: [n for n in tspec if TS.re_dec[pre].match(n)] for pre in "sr"}
    assert len(dec["s"]) == 12 and len(dec["r"]) == 12
    covered = sum(D.get_parameter(n).numel() for n in dec["s"] + dec["r"])
    assert covered > 0.9 * sum(D.get_parameter(n).numel() for n in tspec)        # (the rest: the 1x1 up-convs, heads, biases)
    # every name the rules select is a spectral-norm layer the backward pass corrects (discriminator_path.SN_INDEX)
    for n in sh_w + dec["s"] + dec["r"]:
        assert n[:-len(".weight_orig")] in DP.SN_INDEX
    sizes = [D.get_parameter(n).numel() for n in shared]
    tail = TS.DStepTape._low_tail(shared, sizes)
    assert tail is not None
    first_low = next(i for i, n in enumerate(shared) if n.startswith(f"conv{DP.FLUSH_LEVEL}1"))
    assert tail == sum(sizes[:first_low]) and (sum(sizes) - tail) / sum(sizes) > 0.9


def test_which_layers_take_the_32_channel_winograd_kernel():
    """kernels.winograd_takes is the host's mirror of mtd_conv_winograd_ok plus the size thresholds (no launch, no GPU): the
    generator's 32 -> 32 channel 3x3 layers go to the Winograd kernels on whole-slice maps (side >= WINO_C32_MIN_HW = 128:
    csrc/conv_wino_c32.h) and stay on the halo-tile kernels on the 64 x 64 training patches; the residual-after-activation
    epilogue (MTD_ACT_RELU_ADD) is that form's only; 64-multiples of channels are unaffected by the threshold; a second output
    (the fused masked cotangent) goes there only in the persistent kernel's masked two-output form."""
    from mtd_gan_amd import kernels as K
    assert K.WINO_C32_MIN_HW == 128
    fwd = lambda side: K.geom_fwd(2, side, side, 3, 1, 1)
    for side, want in ((64, False), (128, True), (256, True), (512, True)):
        assert K.winograd_takes(fwd(side), 32, 32, {}) is want, side
        assert K.winograd_takes(K.geom_dgrad_s1(2, side, side, 3, 1), 32, 32, {"act": K.ACT_RELU_ADD}) is want, side
    # ... except where the caller opts in (the generator's forward pass and, since round 6, the data gradients of its plain layers:
    # conv(..., wino32=True)) and the persistent kernel itself takes the launch: one residual operand at most, no scale; a mask and a
    # second output only in its masked two-output form (WINO_C32_BWD), a second output never without a mask
    assert K.WINO_C32_FWD and K.winograd_takes(fwd(64), 32, 32, {"wino32": True, "add1": object(), "act": K.ACT_RELU})
    assert K.WINO_C32_BWD and K.winograd_takes(fwd(64), 32, 32, {"wino32": True, "mask": object()})
    assert K.winograd_takes(fwd(64), 32, 32, {"wino32": True, "mask": object(), "out2": object(), "add1": object()})
    assert not K.winograd_takes(fwd(64), 32, 32, {"wino32": True, "out2": object()})
    assert not K.winograd_takes(fwd(64), 32, 32, {"mask": object(), "out2": object()})                      # (not without the caller's opt-in)
    assert not K.winograd_takes(fwd(64), 32, 32, {"wino32": True, "add2": object()})
    assert not K.winograd_takes(fwd(8), 32, 32, {"wino32": True})
    assert K.winograd_takes(fwd(64), 64, 64, {}) and not K.winograd_takes(fwd(64), 64, 64, {"act": K.ACT_RELU_ADD})
    assert not K.winograd_takes(fwd(512), 32, 32, {"out2": object()})
    assert not K.winograd_takes(fwd(512), 32, 64, {}) and not K.winograd_takes(fwd(512), 64, 32, {})       # (C, N) = (32, 32) only
    assert not K.winograd_takes(K.geom_fwd(2, 130, 130, 3, 1, 1), 32, 32, {})                               # width not a multiple of 4
    assert not K.winograd_takes(K.geom_fwd(2, 512, 512, 4, 2, 1), 32, 32, {})                               # 3x3 stride 1 only


def test_the_switchboard_is_frozen(built_lib, monkeypatch):
    """Round 5: the shipped library reads NO environment variable (its lab switches exist only in an -DMTD_LAB build), run-time
    options go through mtd_set_option by name, and the Python package reads the environment in one place (_options.py: eight
    product switches; everything else only under MTD_LAB=1)."""
    import subprocess
    from mtd_gan_amd import _lib, _options
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und, [ln for ln in und.splitlines() if "getenv" in ln]
    built_lib.mtd_lab_build.restype = ctypes.c_int
    assert built_lib.mtd_lab_build() == 0
    built_lib.mtd_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
    built_lib.mtd_get_option.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    v = ctypes.c_int(-7)
    assert built_lib.mtd_get_option(b"c32f_safe_wait", ctypes.byref(v)) == 0 and v.value == 0
    assert built_lib.mtd_set_option(b"c32f_safe_wait", 1) == 0
    assert built_lib.mtd_get_option(b"c32f_safe_wait", ctypes.byref(v)) == 0 and v.value == 1
    assert built_lib.mtd_set_option(b"c32f_safe_wait", 0) == 0
    assert built_lib.mtd_set_option(b"wino_splitk", 3) == -1 and built_lib.mtd_get_option(None, ctypes.byref(v)) == -1
    # the sources: getenv only inside mtd_lab_env, os.environ only in _options.py / _lib.py (MTD_LAB) / _build.py (HIPCC, MTD_LAB_BUILD)
    csrc = os.path.join(ROOT, "mtd-gan_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            for ln in open(os.path.join(csrc, fn)).read().splitlines():
                if re.search(r"(?<![a-z_])getenv\(", ln):
                    assert fn == "common.h" and "mtd_lab_env" in ln, (fn, ln)
    pkg = os.path.join(ROOT, "mtd-gan_amd")
    for dp_, _d, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py") and fn not in ("_options.py", "_lib.py", "_build.py"):
                assert "os.environ" not in open(os.path.join(dp_, fn)).read(), fn
    assert len(_options.PRODUCT) <= 10
    monkeypatch.setenv("MTD_WINOGRAD", "0")
    assert _options.lab("MTD_WINOGRAD", "1") == ("0" if _options.LAB else "1")
    with pytest.raises(AssertionError):
        _options.product("MTD_WINOGRAD", "1")


def test_product_switches_are_read_from_the_environment():
    """The eight product variables of mtd-gan_amd/_options.py, each through the code that reads it (fresh interpreters: most are read
    at import).  MTD_FORCE_DP and MTD_GRAPH have GPU tests of their own (tests/test_step_gpu.py, tests/test_generator_gpu.py)."""
    import subprocess
    import sys

    def run(code, **env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("MTD_")}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout.strip().splitlines()[-1]
    probe = ("from mtd_gan_amd import train_step as TS, _options as O\n"
             "print(TS.LIST_MODE, TS.LIST_UNDER_DP, O.LAB, O.lab('MTD_WINOGRAD', '1'))")
    assert run(probe) == "True True False 1"
    assert run(probe, MTD_LIST="0", MTD_WINOGRAD="0") == "False True False 1"               # a lab variable without MTD_LAB changes nothing
    assert run(probe, MTD_LIST_DP="0", MTD_LAB="1", MTD_WINOGRAD="0") == "True False True 0"
    gc_probe = ("import gc\nfrom mtd_gan_amd import engine\nengine.freeze_long_lived_objects()\nprint(engine._gc_frozen[0], gc.get_freeze_count() > 0)")
    assert run(gc_probe) == "False False"
    assert run(gc_probe, MTD_GC_FREEZE="1") == "True True"
    wl_probe = "import bench\nprint(bench.parse([]).workload)"
    assert run(wl_probe) == "auto" and run(wl_probe, MTD_BENCH_WORKLOAD="generator") == "generator"
    # MTD_DP_SHARE_GPU: read by bench.py's ranks; the launcher passes it through (two dry-run ranks come up with it set)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e["MTD_DP_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"],
                       env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and '"n_gpus": 2' in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_live_pmc_lookup_and_refusals(monkeypatch):
    """bench.py's own PMC passes (round 5): the per-kernel arithmetic -- dispatch-weighted over every instantiation of a family,
    one symbol for a name with template arguments, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, MFMA
    utilisation = busy cycles / (32 x SQ busy cycles) -- and the cases in which no child process is started at all."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    k6 = "void (anonymous namespace)::wino_conv_kernel<2, false, 6>((anonymous namespace)::WinoParams)"
    k4 = "void (anonymous namespace)::wino_conv_kernel<2, false, 4>((anonymous namespace)::WinoParams)"
    data = {"workload": "full_step", "seconds": 1.0, "sums": {
        "fetch": {(k6, "FETCH_SIZE"): [3000.0, 3], (k4, "FETCH_SIZE"): [100.0, 1]},
        "write": {(k6, "WRITE_SIZE"): [600.0, 3], (k4, "WRITE_SIZE"): [50.0, 1]},
        "mfma": {(k6, "SQ_VALU_MFMA_BUSY_CYCLES"): [1600.0, 3], (k6, "SQ_BUSY_CYCLES"): [100.0, 3]}}}
    traffic, util, prov = bench.live_pmc_lookup(data, "wino_conv_kernel<2, false, 6>")
    assert traffic == round((2 * 1000.0 + 200.0) * 1024) and util == 0.5 and prov["dispatches"] == {"fetch": 3, "write": 3, "mfma": 3}
    assert "live" in prov and prov["source_hash"] == bench._kernel_source_hash()
    t_family, _u, p_family = bench.live_pmc_lookup(data, "wino_conv_kernel")           # the family: both instantiations, launch-weighted
    assert t_family == round((2 * 3100.0 / 4 + 650.0 / 4) * 1024) and p_family["dispatches"]["fetch"] == 4
    assert bench.live_pmc_lookup(data, "igemm_kernel<1, 1, 4, 1>") is None and bench.live_pmc_lookup(None, "x") is None
    roof = {"kernel": "wino_conv_kernel<2, false, 6>", "traffic": 7, "mfma_util_pmc": 0.1, "pmc_provenance": {"files": {}}}
    bench.apply_live_pmc(roof, data)
    assert roof["traffic"] == traffic and roof["mfma_util_pmc"] == 0.5 and roof["committed_pmc"]["traffic"] == 7
    untouched = {"kernel": "igemm_kernel<1, 1, 4, 1>", "traffic": 7}
    bench.apply_live_pmc(untouched, data)
    assert untouched == {"kernel": "igemm_kernel<1, 1, 4, 1>", "traffic": 7}
    # no rocprofv3 on the path / this process under a profiler: nothing is started
    started = []
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: started.append(a) or (_ for _ in ()).throw(AssertionError("child started")))
    monkeypatch.setenv("ROCPROFILER_REGISTER_ROOT", "/x")
    assert bench.live_pmc_collect("full_step") is None and not started
    monkeypatch.delenv("ROCPROFILER_REGISTER_ROOT")
    import shutil
    monkeypatch.setattr(shutil, "which", lambda _n: None)
    monkeypatch.setattr(os.path, "exists", lambda p_, _e=os.path.exists: False if p_ == "/opt/rocm/bin/rocprofv3" else _e(p_))
    assert bench.live_pmc_collect("full_step") is None and not started
    assert bench.parse(["--no-live-pmc"]).no_live_pmc and not bench.parse([]).no_live_pmc


def test_lockstep_driver_groups_aligned_launches(monkeypatch):
    """discriminator_path.disc_backward / disc_backward_lockstep (round 6) without a GPU: a backward pass is a generator that hands out
    its data-gradient launches; run by itself every launch is issued at once, in order; two or three passes advanced together send the
    launches they hand out at the same step as ONE group (kernels.conv_group), a pass that has more to hand out than the others issues
    the rest by itself, everything a pass issues on its own stays in that pass's order, and every pass's return value comes back."""
    from mtd_gan_amd import discriminator_path as DPm
    from mtd_gan_amd import kernels as K
    log = []

    def fake_gen(tag, n, lane=0, **_kw):
        for i in range(n):
            log.append(("own", tag, i))                # (what a pass issues itself between two hand-outs)
            yield ((tag, i), {"lane": lane})
        return "gin-" + tag
    monkeypatch.setattr(DPm, "_disc_backward_gen", lambda tag, n, **kw: fake_gen(tag, n, **kw))
    monkeypatch.setattr(K, "conv", lambda *a, **kw: log.append(("conv", a, kw["lane"])))
    monkeypatch.setattr(K, "conv_group", lambda calls: log.append(("group", [c[0] for c in calls], [c[1]["lane"] for c in calls])))
    assert DPm.disc_backward("a", 2) == "gin-a"
    assert log == [("own", "a", 0), ("conv", ("a", 0), 0), ("own", "a", 1), ("conv", ("a", 1), 0)]
    del log[:]
    res = DPm.disc_backward_lockstep([(("a", 2), {}), (("b", 3), {})])
    assert res == ["gin-a", "gin-b"]
    assert [e for e in log if e[0] != "own"] == [("group", [("a", 0), ("b", 0)], [0, 1]), ("group", [("a", 1), ("b", 1)], [0, 1]),
                                                  ("conv", ("b", 2), 1)]
    assert [e for e in log if e[0] == "own" and e[1] == "b"] == [("own", "b", 0), ("own", "b", 1), ("own", "b", 2)]
    del log[:]
    res = DPm.disc_backward_lockstep([(("a", 1), {}), (("b", 1), {}), (("c", 2), {})])
    assert res == ["gin-a", "gin-b", "gin-c"]
    assert [e for e in log if e[0] != "own"] == [("group", [("a", 0), ("b", 0), ("c", 0)], [0, 1, 2]), ("conv", ("c", 1), 2)]
    # lead: the first pass issues its first hand-outs by itself; a pass given as a callable is built only when it starts -- after them
    del log[:]
    built = []
    res = DPm.disc_backward_lockstep([(("a", 3), {}), lambda: (built.append(len(log)) or (("b", 2), {}))], lead=2)
    assert res == ["gin-a", "gin-b"]
    assert [e for e in log if e[0] != "own"] == [("conv", ("a", 0), 0), ("conv", ("a", 1), 0), ("group", [("a", 2), ("b", 0)], [0, 1]),
                                                  ("conv", ("b", 1), 1)]
    assert built == [log.index(("own", "a", 2)) + 1]      # (built after the lead's launches and what the first pass issued behind them)
    del log[:]
    res = DPm.disc_backward_lockstep([(("a", 1), {}), (("b", 1), {})], lead=5)      # (a lead longer than the pass: it runs alone to its end)
    assert res == ["gin-a", "gin-b"] and [e for e in log if e[0] != "own"] == [("conv", ("a", 0), 0), ("conv", ("b", 0), 1)]
