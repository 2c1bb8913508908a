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
