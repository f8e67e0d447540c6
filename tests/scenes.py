"""Scene files shared by the CPU tests (oracle against the reference's own kernels) and the GPU tests (HIP path against
the oracle and against the reference's kernels): written to a temporary directory, loaded through the host library."""
import numpy as np


def write_thin_lens_hdr_scene(tmp_path):
    """A thin-lens camera (aperture sampling in kernel_generate, CUDA/Camera.h:20-62), an HDR environment map with
    structure (sample_sky on every miss, CUDA/Sky.h:7-16; a Radiance .hdr file written here), and eight instances of a
    file mesh with rotation + uniform scale, half of them rough plastic, over a diffuse ground plane; no emitters:
    the sky lights the scene. Returns (scene xml, sky file)."""
    w, h = 32, 16
    yy, xx = np.mgrid[0:h, 0:w]
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[:, :, 0] = 100 + 100 * np.sin(xx / 5.0); rgbe[:, :, 1] = 120 + 80 * np.cos(yy / 3.0); rgbe[:, :, 2] = 60 + 3 * xx; rgbe[:, :, 3] = 128 + (yy < 6) * 2
    rows = b"".join(bytes([2, 2, 0, w]) + b"".join(b"".join(bytes([1, int(v)]) for v in rgbe[y, :, c]) for c in range(4)) for y in range(h))
    (tmp_path / "sky.hdr").write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + rows)
    (tmp_path / "pyramid.obj").write_text("v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nv 0 1.5 0\nf 1 2 5\nf 2 3 5\nf 3 4 5\nf 4 1 5\nf 1 3 2\nf 1 4 3\n")
    rng = np.random.default_rng(1)
    xml = ('<scene version="0.5.0"><integrator type="path"><integer name="maxDepth" value="4"/></integrator>'
           '<sensor type="thinlens"><float name="fov" value="45"/><float name="apertureRadius" value="0.15"/><float name="focusDistance" value="4"/>'
           '<transform name="toWorld"><lookat origin="0, 2.5, 7" target="0, 0.5, 0" up="0, 1, 0"/></transform></sensor>'
           '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="-90"/><scale value="6"/></transform><bsdf type="diffuse"><rgb name="reflectance" value="0.7, 0.6, 0.5"/></bsdf></shape>')
    for i in range(8):
        xml += ('<shape type="obj"><string name="filename" value="pyramid.obj"/><transform name="toWorld"><scale value="%.2f"/><rotate y="1" angle="%.1f"/><translate x="%.2f" y="0.01" z="%.2f"/></transform>'
                % (rng.uniform(0.4, 1.2), rng.uniform(0, 360), rng.uniform(-4, 4), rng.uniform(-3, 2)))
        xml += ('<bsdf type="roughplastic"><rgb name="diffuseReflectance" value="0.3, 0.5, 0.8"/><float name="alpha" value="0.2"/></bsdf></shape>' if i % 2 else
                '<bsdf type="diffuse"><rgb name="reflectance" value="0.8, 0.3, 0.2"/></bsdf></shape>')
    (tmp_path / "s.xml").write_text(xml + "</scene>")
    return str(tmp_path / "s.xml"), str(tmp_path / "sky.hdr")


def write_scene_with_everything(tmp_path, png_bytes):
    """Every feature at once: a textured rough-plastic floor (uv repeat, mip maps), two emitters of different power of
    which one is a rotated, scaled file mesh (light_mesh_transform_indices), a rough dielectric holding a
    back-scattering medium, a named conductor. `png_bytes(pixels, colour_type, depth)` encodes the texture."""
    rng = np.random.default_rng(2)
    (tmp_path / "t.png").write_bytes(png_bytes(rng.integers(0, 256, (32, 32, 3)), 2, 8))
    (tmp_path / "quad.obj").write_text("v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nvt 0 0\nvt 3 0\nvt 3 3\nvt 0 3\nf 1/1 2/2 3/3\nf 1/1 3/3 4/4\n")
    (tmp_path / "s.xml").write_text(
        '<scene version="0.5.0"><integrator type="path"><integer name="maxDepth" value="6"/></integrator>'
        '<sensor type="perspective"><float name="fov" value="50"/><transform name="toWorld"><lookat origin="0, 2, 6" target="0, 0.7, 0" up="0, 1, 0"/></transform></sensor>'
        '<shape type="obj"><string name="filename" value="quad.obj"/><transform name="toWorld"><scale value="5"/></transform><bsdf type="roughplastic"><texture type="bitmap" name="diffuseReflectance"><string name="filename" value="t.png"/></texture><float name="alpha" value="0.25"/></bsdf></shape>'
        '<shape type="obj"><string name="filename" value="quad.obj"/><transform name="toWorld"><scale value="0.5"/><rotate x="1" angle="180"/><rotate z="1" angle="20"/><translate x="1.5" y="2.5" z="0.25"/></transform><emitter type="area"><rgb name="radiance" value="9, 8, 7"/></emitter></shape>'
        '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="90"/><scale value="0.4"/><translate x="-1.5" y="3"/></transform><emitter type="area"><rgb name="radiance" value="20, 5, 5"/></emitter></shape>'
        '<shape type="sphere"><float name="radius" value="0.7"/><transform name="toWorld"><translate y="0.7"/></transform><bsdf type="roughdielectric"><float name="intIOR" value="1.5"/><float name="alpha" value="0.2"/></bsdf>'
        '<medium type="homogeneous" name="interior"><rgb name="sigmaA" value="0.5, 0.2, 0.1"/><rgb name="sigmaS" value="2, 2.5, 3"/><phase type="hg"><float name="g" value="-0.4"/></phase></medium></shape>'
        '<shape type="sphere"><float name="radius" value="0.4"/><transform name="toWorld"><translate x="-1.6" y="0.4" z="1"/></transform><bsdf type="roughconductor"><string name="material" value="Cu"/><float name="alpha" value="0.05"/></bsdf></shape></scene>')
    return str(tmp_path / "s.xml")
