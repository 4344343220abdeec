"""YAML scene loader: the reference's serde schema (scene.rs:40-74, shapes.rs:15-81)."""
import copy
import os

import pytest
import yaml

from conftest import SCENES


def test_demo_scenes_load(flux, demo1, demo2):
    assert demo1.scene_name == "demo1" and len(demo1.shapes) == 6
    assert demo2.scene_name == "demo2" and len(demo2.shapes) == 13
    assert demo2.output_settings.image_width == 800 and demo2.output_settings.image_height == 600
    assert demo2.camera_data.lens_radius == 0.09 and demo1.camera_data.lens_radius == 0.0
    s0 = demo2.shapes[0]
    assert isinstance(s0, flux.SphereData) and s0.invert and s0.radius == 100.0
    assert isinstance(s0.material, flux.EmissiveData) and s0.material.power == 0.3
    # anchors/aliases (&mat1 / *mat1) resolve; unknown top-level keys mat1..3 are ignored
    m = demo2.shapes[2].material
    assert isinstance(m, flux.GlossyReflectiveData) and m.reflect_exponent == 10000.0
    assert demo2.shapes[5].material == m and demo2.shapes[3].material.reflect_exponent == 100.0
    p = demo2.shapes[12]
    assert isinstance(p, flux.PlaneData) and p.normal == (0.0, 1.0, 0.0)
    assert isinstance(p.material, flux.MatteData) and p.material.diffuse_coefficient == 1.0
    # ints are accepted where the schema says f64 (serde_yaml coerces)
    assert demo1.camera_settings.look_at == (2.5, 1.0, 0.0)


def _doc():
    return yaml.safe_load(open(os.path.join(SCENES, "demo1.yml")))


def test_missing_field_is_an_error(flux):
    d = _doc()
    del d["camera_data"]["lens_radius"]
    with pytest.raises(flux.SceneError, match="lens_radius"):
        flux.scene_from_dict(d)
    d = _doc()
    del d["shapes"][1]["Sphere"]["invert"]
    with pytest.raises(flux.SceneError, match="invert"):
        flux.scene_from_dict(d)
    d = _doc()
    del d["background"]
    with pytest.raises(flux.SceneError, match="background"):
        flux.scene_from_dict(d)


def test_unknown_variant_is_an_error(flux):
    d = _doc()
    d["shapes"][0] = {"Torus": {}}
    with pytest.raises(flux.SceneError, match="unknown variant `Torus`"):
        flux.scene_from_dict(d)
    d = _doc()
    d["shapes"][1]["Sphere"]["material"] = {"Glass": {}}
    with pytest.raises(flux.SceneError, match="unknown variant `Glass`"):
        flux.scene_from_dict(d)


def test_bad_types_are_errors(flux):
    d = _doc()
    d["camera_settings"]["eye"] = [1, 2]
    with pytest.raises(flux.SceneError):
        flux.scene_from_dict(d)
    d = _doc()
    d["output_settings"]["image_width"] = -3
    with pytest.raises(flux.SceneError):
        flux.scene_from_dict(d)
    d = _doc()
    d["shapes"][1]["Sphere"]["radius"] = "big"
    with pytest.raises(flux.SceneError):
        flux.scene_from_dict(d)


def test_unknown_fields_are_ignored(flux):
    d = _doc()
    d["extra_top_level"] = 1
    d["shapes"][1]["Sphere"]["comment"] = "ignored"
    sd = flux.scene_from_dict(d)
    assert len(sd.shapes) == 6


def test_flatten_to_abi(flux, demo2):
    from flux_amd.scene import SceneDesc
    sdesc = SceneDesc(demo2)
    d = sdesc.desc
    assert d.num_shapes == 13 and d.image_width == 800 and d.scene_name == b"demo2"
    s1 = sdesc.shapes[1]
    assert s1.kind == flux._lib.SHAPE_SPHERE and list(s1.p) == [-9.0, 7.0, 8.0] and s1.radius == 5.0
    assert s1.material.kind == flux._lib.MAT_EMISSIVE and s1.material.k == 10.0
    g = sdesc.shapes[2].material
    assert g.kind == flux._lib.MAT_GLOSSY and g.k == 0.5 and g.exponent == 10000.0 and list(g.color) == [0.8, 0.6, 1.0]
    pl = sdesc.shapes[12]
    assert pl.kind == flux._lib.SHAPE_PLANE and list(pl.n) == [0.0, 1.0, 0.0]
    assert pl.material.kind == flux._lib.MAT_MATTE and list(pl.material.ambient) == [1.0, 1.0, 1.0]
    assert sdesc.shapes[0].invert == 1
