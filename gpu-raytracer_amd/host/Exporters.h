// Image writers behind `-o file.ppm|.exr` (reference: Src/Exporters/PPMExporter.cpp:7-39,
// EXRExporter.cpp:10-59, Main.cpp:199-246). `data` is the frame as the integrator holds it:
// RGB per pixel at x + y * pitch with row 0 at the BOTTOM; both writers flip it.
#pragma once
#include <string>
#include <vector>

#include "Math.h"

namespace PPMExporter {
	// 8-bit binary PPM of values already in display space (see Exporters::tonemap)
	bool save(const std::string & filename, int pitch, int width, int height, const std::vector<Vector3> & data);
}

namespace EXRExporter {
	// Scan-line OpenEXR, channels B G R stored as 16-bit half, no compression -- what the
	// reference's tinyexr call produces for its zero-initialised header
	bool save(const std::string & filename, int pitch, int width, int height, const std::vector<Vector3> & data);
}

namespace Exporters {
	// What the reference's window does to the frame before an LDR screenshot is read back
	// (Shaders/post.frag:14-35, Window.cpp:164-184): clamp below at 0, ACES filmic curve, gamma
	// 1/2.2, then the round trip through the 8-bit back buffer.
	Vector3 tonemap(Vector3 colour);

	// Picks the writer by extension (.ppm: tone-mapped LDR, .exr: raw radiance); false + message for others
	bool save(const std::string & filename, int pitch, int width, int height, const std::vector<Vector3> & radiance, std::string * error);
}
