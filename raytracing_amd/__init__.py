"""MI355X-native path-tracing core: Python face of librt_mi355x.so (HIP kernels behind a C ABI)."""
from .renderer import Group, Renderer, RtError, SceneData, build_scene, load_scene  # noqa: F401
from . import scenes  # noqa: F401
from . import controls  # noqa: F401
from .controls import Camera, HeadlessApp  # noqa: F401
