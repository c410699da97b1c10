import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the CPU-side libraries on demand (the HIP library is built by __graft_entry__.build())."""
    import subprocess
    need = [os.path.join(ROOT, "basicrenderer_amd", "lib", "libbrmi_scene.so"), os.path.join(ROOT, "oracle", "_build", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "scene", "oracle"], stdout=subprocess.DEVNULL)
    # the reference's LOD builder is compiled from the reference checkout where that exists (this container); the built
    # library travels to the GPU box with the snapshot
    ref = os.path.join(ROOT, "oracle", "_ref", "libclodref.so")
    if not os.path.exists(ref) and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref")], stdout=subprocess.DEVNULL)
    yield


def have_clodref():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libclodref.so"))


def Scene(*args, **kw):
    """basicrenderer_amd.Scene for tests: lod_builder="clusterlod" runs the meshes through the REFERENCE's clodBuild, which only the
    test side may load (tests/clodref_bridge.py) and hands to the scene library as a caller-supplied DAG builder."""
    from basicrenderer_amd import Scene as ProductScene
    if kw.get("lod_builder") == "clusterlod":
        import clodref_bridge
        kw = dict(kw)
        del kw["lod_builder"]
        kw["dag_builder"] = clodref_bridge.dag_builder()
    return ProductScene(*args, **kw)


SCENE_CASES = {
    # name: (preset, W, H, kwargs)
    "tiny": ("tiny", 256, 144, dict(point_lights=6)),
    "tiny_lod": ("tiny", 320, 180, dict(point_lights=3, lod_levels=2)),
    "tiny_coat_fuzz": ("tiny", 256, 144, dict(point_lights=6, material_features=3)),
    "sponza_coat_fuzz": ("sponza", 480, 270, dict(point_lights=32, size_scale=0.15, material_features=3)),
    "sponza_small": ("sponza", 640, 360, dict(point_lights=64, size_scale=0.25)),
    "bistro_small": ("bistro", 640, 360, dict(point_lights=256, size_scale=0.08)),
    # compute skinning: bone-merged meshlet bounds in the cull, skinned vertices in raster and resolve
    "tiny_skinned": ("tiny", 256, 144, dict(point_lights=4, skinned_fraction=1.0, lod_levels=2)),
    "bistro_skinned": ("bistro", 640, 360, dict(point_lights=32, size_scale=0.3, skinned_fraction=0.3)),
    # LOD DAG from the reference's own builder (meshoptimizer + clusterlod.h, oracle/_ref): irregular meshlets, ~400-cluster groups
    # spot lights (cone attenuation in the shader, cone bounding spheres in the light clustering)
    "sponza_spots": ("sponza", 640, 360, dict(point_lights=48, size_scale=0.25, spot_every=2, material_features=1)),
    # mirrored instances drawn with reversed winding
    "bistro_mirrored": ("bistro", 640, 360, dict(point_lights=16, size_scale=0.3, material_features=4)),
    # UV streams + material textures through the software sampler (base colour, metallic / roughness, AO, emissive, normal map)
    "tiny_textured": ("tiny", 256, 144, dict(point_lights=6, lod_levels=2, material_features=8)),
    "sponza_textured": ("sponza", 640, 360, dict(point_lights=32, size_scale=0.25, material_features=8 | 3)),
    # alpha-tested materials: per-pixel texcoord + SampleLevel in the rasteriser (direct, binned and overflow paths)
    "tiny_alpha": ("tiny", 256, 144, dict(point_lights=6, lod_levels=2, material_features=24)),
    "sponza_alpha": ("sponza", 640, 360, dict(point_lights=32, size_scale=0.25, material_features=24)),
    "bistro_alpha_skinned": ("bistro", 640, 360, dict(point_lights=32, size_scale=0.3, material_features=24 | 4, skinned_fraction=0.3)),
    # RGBA8 vertex colours in the pages tint the base colour (constant-factor and textured materials)
    "tiny_vcolor": ("tiny", 256, 144, dict(point_lights=6, lod_levels=2, material_features=32)),
    "sponza_vcolor_textured": ("sponza", 640, 360, dict(point_lights=32, size_scale=0.25, material_features=32 | 24 | 3)),
    # OpenPBR coat / fuzz texture slots (ApplyOpenPBRTextureSampling) on layered materials
    "sponza_layer_textures": ("sponza", 640, 360, dict(point_lights=32, size_scale=0.25, material_features=64 | 8 | 3)),
    "tiny_layer_textures_only": ("tiny", 256, 144, dict(point_lights=4, lod_levels=2, material_features=64 | 16 | 3 | 32)),
    # contact-refinement parallax: the height map moves the texcoord of every slot (with / without a normal map, rays that never hit)
    "tiny_parallax": ("tiny", 256, 144, dict(point_lights=6, lod_levels=2, material_features=128 | 8)),
    "sponza_parallax_all": ("sponza", 640, 360, dict(point_lights=32, size_scale=0.25, material_features=128 | 64 | 32 | 24 | 3)),
    # LOD DAG from this library's own cluster-LOD builder (lod_builder.cpp)
    "tiny_ownlod": ("tiny", 256, 144, dict(point_lights=4, lod_builder="own")),
    "sponza_ownlod": ("sponza", 640, 360, dict(point_lights=32, size_scale=0.25, lod_builder="own")),
    "bistro_ownlod_skinned": ("bistro", 640, 360, dict(point_lights=32, size_scale=0.3, skinned_fraction=0.3, lod_builder="own")),
    "sponza_ownlod_alpha": ("sponza", 640, 360, dict(point_lights=16, size_scale=0.25, lod_builder="own", material_features=24)),
    "tiny_clod": ("tiny", 256, 144, dict(point_lights=4, lod_builder="clusterlod")),
    "sponza_clod": ("sponza", 640, 360, dict(point_lights=32, size_scale=0.25, lod_builder="clusterlod")),
    "bistro_clod_skinned": ("bistro", 640, 360, dict(point_lights=32, size_scale=0.3, skinned_fraction=0.3, lod_builder="clusterlod")),
    "sponza_clod_alpha": ("sponza", 640, 360, dict(point_lights=16, size_scale=0.25, lod_builder="clusterlod", material_features=24)),
}


@pytest.fixture(scope="session")
def scenes():
    cache = {}

    def get(name):
        if name not in cache:
            preset, W, H, kw = SCENE_CASES[name]
            cache[name] = Scene(preset, W, H, **kw)
        return cache[name]

    return get


@pytest.fixture(scope="session")
def oracle_frames(scenes):
    import orc
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = orc.OracleFrame(scenes(name)).run()
        return cache[name]

    return get


def caller_mesh_scene(width=480, height=270, **kw):
    """A scene of meshes this file makes up (not the generator's patches) through brmi_scene_create_from_meshes: a torus with a UV seam, an
    open height field without normals (derived by the library) and a fan of large triangles; two instances of the torus, one mirrored."""
    from basicrenderer_amd import Scene as RawScene

    def torus(nu=96, nv=48, R=1.0, r=0.4):
        u = np.linspace(0, 2 * np.pi, nu, endpoint=False); v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
        U, V = np.meshgrid(u, v, indexing="ij")
        P = np.stack([(R + r * np.cos(V)) * np.cos(U), r * np.sin(V), (R + r * np.cos(V)) * np.sin(U)], -1).reshape(-1, 3)
        N = np.stack([np.cos(V) * np.cos(U), np.sin(V), np.cos(V) * np.sin(U)], -1).reshape(-1, 3)
        uv = np.stack([U / (2 * np.pi) * 4, V / (2 * np.pi) * 2], -1).reshape(-1, 2)
        i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
        a = (i * nv + j).ravel(); b = (((i + 1) % nu) * nv + j).ravel(); c = (((i + 1) % nu) * nv + (j + 1) % nv).ravel(); d = (i * nv + (j + 1) % nv).ravel()
        return dict(positions=P.astype(np.float32), normals=N.astype(np.float32), uvs=uv.astype(np.float32), indices=np.stack([a, c, b, a, d, c], 1).ravel().astype(np.uint32), material=0)

    def field(n=80):
        x, z = np.meshgrid(np.linspace(-4, 4, n + 1), np.linspace(-4, 4, n + 1), indexing="ij")
        y = -0.8 + 0.15 * np.sin(x * 2.1) * np.cos(z * 1.7) + 0.04 * np.sin(x * 9.0 + z * 7.0)
        P = np.stack([x, y, z], -1).reshape(-1, 3)
        i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        a = (i * (n + 1) + j).ravel(); b = a + (n + 1); c = b + 1; d = a + 1
        return dict(positions=P.astype(np.float32), indices=np.stack([a, d, c, a, c, b], 1).ravel().astype(np.uint32), uvs=(P[:, [0, 2]] * 0.5).astype(np.float32), material=1)

    def fan():
        P = np.array([[0, 3.0, -6.0]] + [[8 * np.cos(t), 3.0 + 4 * np.sin(t), -6.0] for t in np.linspace(0, 2 * np.pi, 13)[:-1]], dtype=np.float32)
        I = np.array([[0, k + 1, (k + 1) % 12 + 1] for k in range(12)], dtype=np.uint32).ravel()
        return dict(positions=P, indices=I, material=2)

    eye = np.eye(4, dtype=np.float32)
    moved = eye.copy(); moved[3, :3] = [2.4, 0.2, -0.8]
    mirrored = np.diag([-0.7, 0.7, 0.7, 1.0]).astype(np.float32); mirrored[3, :3] = [-2.2, 0.4, 0.3]
    kw.setdefault("point_lights", 6)
    return RawScene(width=width, height=height, meshes=[torus(), field(), fan()],
                    instances=[(0, eye), (0, moved), (0, mirrored, True), (1, eye), (2, eye)], view=dict(eye=(0.5, 1.6, 4.5), yaw=0.05, pitch=-0.3, fov=65, near=0.1, far=200), **kw)
