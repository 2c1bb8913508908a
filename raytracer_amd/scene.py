"""Scene and camera data of the SPPM path, as plain tables.

The values are the reference's Cornell-box sphere scene (/root/reference/src/scene.h:12-24) and the
camera of src/sppm2.h:48-56.  In the reference the scene is a const global compiled into every
translation unit; here it is run-time data handed to the C-ABI as an array of
`sppm_sphere` (include/sppm_hip.h).
"""
import math

DIFFUSE, SPECULAR, REFRACTION = 0, 1, 2  # edupt::ReflectionType, src/material.h:10-15

# (radius, position, emission, color, reflection_type); sphere 0 is the light (LightID, src/sppm2.h:132)
CORNELL_SPHERES = [
    (5.0, (50.0, 75.0, 81.6), (36.0, 36.0, 36.0), (0.999, 0.999, 0.999), DIFFUSE),
    (1e5, (1e5 + 1, 40.8, 81.6), (0.0, 0.0, 0.0), (0.75, 0.25, 0.25), DIFFUSE),
    (1e5, (-1e5 + 99, 40.8, 81.6), (0.0, 0.0, 0.0), (0.25, 0.25, 0.75), DIFFUSE),
    (1e5, (50.0, 40.8, 1e5), (0.0, 0.0, 0.0), (0.75, 0.75, 0.75), DIFFUSE),
    (1e5, (50.0, 40.8, -1e5 + 250), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), DIFFUSE),
    (1e5, (50.0, 1e5, 81.6), (0.0, 0.0, 0.0), (0.75, 0.75, 0.75), DIFFUSE),
    (1e5, (50.0, -1e5 + 81.6, 81.6), (0.0, 0.0, 0.0), (0.75, 0.75, 0.75), DIFFUSE),
    (20.0, (65.0, 20.0, 20.0), (0.0, 0.0, 0.0), (0.25, 0.75, 0.25), DIFFUSE),
    (16.5, (27.0, 16.5, 47.0), (0.0, 0.0, 0.0), (0.99, 0.99, 0.99), SPECULAR),
    (16.5, (77.0, 16.5, 78.0), (0.0, 0.0, 0.0), (0.99, 0.99, 0.99), REFRACTION),
]

CAMERA_POSITION = (50.0, 52.0, 220.0)
CAMERA_UP = (0.0, 1.0, 0.0)
SCREEN_HEIGHT = 30.0
SCREEN_DIST = 40.0


def camera_direction():
    """normalize(Vec(0, -0.04, -1)) evaluated as src/vec.h:19 does: v * (1.0 / length)."""
    x, y, z = 0.0, -0.04, -1.0
    inv = 1.0 / math.sqrt(x * x + y * y + z * z)
    return (x * inv, y * inv, z * inv)


class _XorShift:
    """edupt::XorShift (src/random.h:9-33), the generator the scene below is placed with."""

    def __init__(self, seed):
        s = seed & 0xFFFFFFFF
        self.s = []
        for i in range(1, 5):
            s = (1812433253 * (s ^ (s >> 30)) + i) & 0xFFFFFFFF
            self.s.append(s)

    def next01(self):
        s = self.s
        t = (s[0] ^ (s[0] << 11)) & 0xFFFFFFFF
        s[0], s[1], s[2] = s[1], s[2], s[3]
        s[3] = ((s[3] ^ (s[3] >> 19)) ^ (t ^ (t >> 8))) & 0xFFFFFFFF
        return s[3] / 4294967295.0


def synthetic_spheres(n, seed=20260517, radius=(0.4, 1.2), mix=(0.5, 0.25, 0.25)):
    """BASELINE config 5's scene (no reference counterpart, SURVEY.md R9: the reference's scene is a 10-sphere
    constant): the Cornell box's light and six walls (spheres 0..6 of src/scene.h:12-24) and n small spheres
    placed inside the box by one XorShift(seed) stream -- per sphere, in this order: x in [4, 96], y in [2, 70],
    z in [4, 166], radius in `radius`, a material draw against `mix` = (diffuse, specular, refraction) shares,
    and for diffuse spheres three colour draws in [0.25, 0.9] (mirror and glass are 0.99 grey like the
    reference's).  Spheres may overlap each other; one that would touch the light is drawn again."""
    rng = _XorShift(seed)
    out = list(CORNELL_SPHERES[:7])
    lx, ly, lz = CORNELL_SPHERES[0][1]
    while len(out) < 7 + n:
        x = 4.0 + 92.0 * rng.next01()
        y = 2.0 + 68.0 * rng.next01()
        z = 4.0 + 162.0 * rng.next01()
        r = radius[0] + (radius[1] - radius[0]) * rng.next01()
        m = rng.next01()
        kind = DIFFUSE if m < mix[0] else (SPECULAR if m < mix[0] + mix[1] else REFRACTION)
        if kind == DIFFUSE:
            col = tuple(0.25 + 0.65 * rng.next01() for _ in range(3))
        else:
            col = (0.99, 0.99, 0.99)
        if math.sqrt((x - lx) ** 2 + (y - ly) ** 2 + (z - lz) ** 2) < CORNELL_SPHERES[0][0] + r + 0.5:
            continue
        out.append((r, (x, y, z), (0.0, 0.0, 0.0), col, kind))
    return out
