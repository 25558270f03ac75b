"""The `icp-settings-*.yaml` surface (replaces load_icp_set_of_params, src/LidarOdometry.cpp:57-88)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARAMS = os.path.join(ROOT, "params")


def test_regular_settings_every_key(pkg):
    p = pkg.Parameters.load_from(open(os.path.join(PARAMS, "icp-settings-regular.yaml")).read())
    assert p.max_iterations == 100
    assert p.min_abs_step_trans == pytest.approx(5e-5) and p.min_abs_step_rot == pytest.approx(1e-5)
    assert p.use_scale_outlier_detector == 1 and p.scale_outlier_threshold == pytest.approx(1.1)
    assert p.use_robust_kernel == 0
    assert p.robust_kernel_param == pytest.approx(np.deg2rad(0.1)) and p.robust_kernel_scale == pytest.approx(400.0)
    assert p.solver_class == pkg._lib.SOLVER_GAUSS_NEWTON and p.solver_max_iterations == 20
    assert p.matcher_class == pkg._lib.MATCHER_POINT2PLANE
    assert p.matcher_threshold == pytest.approx(0.70) and p.plane_eigen_threshold == pytest.approx(0.07)
    assert p.knn == 6 and p.run_from_iteration == 0 and p.run_up_to_iteration == 0
    assert p.quality_class == pkg._lib.QUALITY_PAIRED_RATIO and p.quality_threshold == pytest.approx(0.10)


def test_p2p_horn_settings(pkg):
    p = pkg.Parameters.load_from_file(os.path.join(PARAMS, "icp-settings-p2p-horn.yaml"))
    assert p.solver_class == pkg._lib.SOLVER_HORN
    assert p.matcher_class == pkg._lib.MATCHER_POINTS_DISTANCE_THRESHOLD
    assert p.matcher_threshold == pytest.approx(0.70)
    assert p.nn_kernel == pkg.NN_AUTO and p.fixed_iterations == 0


def test_include_and_mola_dir(pkg):
    # kitti-default.yaml pulls the three ICP cases in through $include{$(mola-dir ...)/...}
    # like params/kitti-default.yaml:43,46,50 of the reference
    f = os.path.join(PARAMS, "kitti-default.yaml")
    a = pkg.Parameters.load_from_file(f, mola_dir=ROOT, key="icp_settings_with_vel")
    b = pkg.Parameters.load_from_file(f, mola_dir=ROOT, key="icp_settings_loop_closure")
    assert a.matcher_class == pkg._lib.MATCHER_POINT2PLANE      # the reference's file: all three cases are its shipped pipeline
    assert b.matcher_class == pkg._lib.MATCHER_POINT2PLANE
    c = pkg.Parameters.load_from_file(os.path.join(PARAMS, "kitti-p2p-horn.yaml"), mola_dir=ROOT, key="icp_settings_with_vel")
    assert c.matcher_class == pkg._lib.MATCHER_POINTS_DISTANCE_THRESHOLD
    with pytest.raises(pkg.IcpError) as e:
        pkg.Parameters.load_from_file(f, mola_dir=None, key="icp_settings_with_vel")
    assert e.value.status == pkg._lib.E_CONFIG and "mola-dir" in str(e.value)
    with pytest.raises(pkg.IcpError):
        pkg.Parameters.load_from_file(f, mola_dir=ROOT, key="no_such_key")


@pytest.mark.skipif(not os.path.isdir("/root/reference/params"), reason="reference tree not present on this box")
def test_reference_files_parse_verbatim(pkg):
    for name in ("icp-settings-regular.yaml", "icp-settings-loop-closure.yaml"):
        p = pkg.Parameters.load_from_file(os.path.join("/root/reference/params", name))
        assert p.max_iterations == 100 and p.knn == 6 and p.quality_threshold == pytest.approx(0.10)
    p = pkg.Parameters.load_from_file("/root/reference/params/kitti-default.yaml", mola_dir="/root/reference",
                                      key="icp_settings_with_vel")
    assert p.matcher_threshold == pytest.approx(0.70)


BASE = """
icp_class: {icp}
params:
  maxIterations: 7
solvers:
  - class: {solver}
matchers:
  - class: {matcher}
    params:
      threshold: 0.25
quality:
  - class: {quality}
"""
GOOD = dict(icp="mp2p_icp::ICP", solver="mp2p_icp::Solver_Horn", matcher="mp2p_icp::Matcher_Points_DistanceThreshold",
            quality="mp2p_icp::QualityEvaluator_PairedRatio")


def test_minimal_document_and_defaults(pkg):
    p = pkg.Parameters.load_from(BASE.format(**GOOD))
    assert p.max_iterations == 7 and p.matcher_threshold == pytest.approx(0.25)
    assert p.quality_threshold == pytest.approx(0.10)  # default


@pytest.mark.parametrize("field,value", [("icp", "mp2p_icp::ICP_Bogus"), ("solver", "mp2p_icp::Solver_Nope"),
                                         ("matcher", "mp2p_icp::Matcher_Nope"),
                                         ("quality", "mp2p_icp::QualityEvaluator_Nope")])
def test_unknown_class_fails_loudly(pkg, field, value):
    # mirrors the hard error at src/LidarOdometry.cpp:70-75: the message names the class
    with pytest.raises(pkg.IcpError) as e:
        pkg.Parameters.load_from(BASE.format(**{**GOOD, field: value}))
    assert e.value.status == pkg._lib.E_CONFIG and value in str(e.value)


@pytest.mark.parametrize("missing", ["icp_class", "params", "solvers", "matchers", "quality"])
def test_required_entries(pkg, missing):
    # ENSURE_YAML_ENTRY_EXISTS at src/LidarOdometry.cpp:63,77,80,83,86
    doc = BASE.format(**GOOD)
    lines, out, skip = doc.splitlines(), [], False
    for ln in lines:
        if ln.startswith(missing):
            skip = True
            continue
        if skip and (ln.startswith(" ") or ln.startswith("-")):
            continue
        skip = False
        out.append(ln)
    with pytest.raises(pkg.IcpError) as e:
        pkg.Parameters.load_from("\n".join(out))
    assert missing in str(e.value)


def test_yaml_syntax_errors(pkg):
    for bad in ("icp_class mp2p_icp::ICP", "a:\n\tb: 1", "icp_class: x\nicp_class: y"):
        with pytest.raises(pkg.IcpError) as e:
            pkg.Parameters.load_from(bad)
        assert e.value.status == pkg._lib.E_CONFIG


def test_shipped_pipeline_needs_plane_stages(pkg):
    # the reference's shipped pipeline (Point2Plane + GaussNewton) parses and is runnable on the HIP stages; caller-
    # supplied point-to-point stages cannot serve it and the loop says so (no silent substitution)
    p = pkg.Parameters.load_from(open(os.path.join(PARAMS, "icp-settings-regular.yaml")).read())
    with pytest.raises(pkg.IcpError) as e:
        pkg.run_loop(lambda T, thr: 0, lambda *a: np.zeros(24), np.eye(4), p, 10, 10)
    assert e.value.status == pkg._lib.E_UNSUPPORTED and "point-to-plane" in str(e.value)
    # inconsistent stage combinations are rejected by name
    q = pkg.Parameters.load_from(open(os.path.join(PARAMS, "icp-settings-regular.yaml")).read())
    q.solver_class = pkg._lib.SOLVER_HORN
    with pytest.raises(pkg.IcpError) as e:
        pkg.run_loop(lambda T, thr: 0, lambda *a: np.zeros(24), np.eye(4), q, 10, 10)
    assert e.value.status == pkg._lib.E_UNSUPPORTED and "Solver_GaussNewton" in str(e.value)
    q = pkg.Parameters.load_from(open(os.path.join(PARAMS, "icp-settings-regular.yaml")).read())
    q.knn = 17     # (3 .. 16 are served since round 5)
    with pytest.raises(pkg.IcpError) as e:
        pkg.run_loop(lambda T, thr: 0, lambda *a: np.zeros(24), np.eye(4), q, 10, 10)
    assert e.value.status == pkg._lib.E_UNSUPPORTED and "knn" in str(e.value)
