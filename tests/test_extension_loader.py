"""``monai_amd._extensions.load_module`` (boundary B2 of SURVEY.md 8b): the reference's loader semantics on hipcc."""
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture()
def loader():
    from monai_amd._extensions import loader as ld

    ld.EXTENSION_DIRS.append(os.path.join(HERE, "ext_sample"))
    try:
        yield ld
    finally:
        ld.EXTENSION_DIRS.pop()


def test_unknown_module_raises_like_the_reference(loader):
    for name in ("gmm", "no_such_extension", "_build"):
        with pytest.raises(ValueError, match="No extension module named"):
            loader.load_module(name)


def test_build_cache_and_defines(loader):
    m1 = loader.load_module("axpb")
    assert os.path.exists(m1.__file__) and m1.axpb_scale() == 1 and hasattr(m1.cdll, "axpb_f32")
    t = os.path.getmtime(m1.__file__)
    assert os.path.getmtime(loader.load_module("axpb").__file__) == t          # second call: the cached artefact
    m3 = loader.load_module("axpb", defines={"AXPB_SCALE": 3})                  # defines -> another artefact, named by the values
    assert m3.__file__ != m1.__file__ and "_3_" in os.path.basename(m3.__file__) and m3.axpb_scale() == 3
    assert "gfx950" in os.path.basename(m1.__file__)


def test_build_timeout(loader):
    with pytest.raises(TimeoutError, match="Build appears to be blocked"):
        loader.load_module("axpb", defines={"AXPB_SCALE": 5, "UNIQUE": os.getpid()}, build_timeout=0.01)


def test_names_outside_the_extension_dirs_are_rejected():
    """a name with a path separator or '..' never leaves EXTENSION_DIRS (the reference only ever joins a bare directory name)"""
    import pytest

    from monai_amd._extensions import load_module

    for bad in ("../csrc", "a/b", "..", os.sep + "tmp"):
        with pytest.raises(ValueError):
            load_module(bad)
