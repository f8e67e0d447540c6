#include "Exporters.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace {
struct File {
	FILE * f;
	File(const std::string & name, const char * mode) : f(fopen(name.c_str(), mode)) { }
	~File() { if (f) fclose(f); }
};

inline float clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// float -> IEEE half the way the reference's tinyexr does it (tinyexr.h float_to_half_full): the first dropped bit
// alone decides the rounding -- ties go away from zero, not to even --, float subnormals flush to zero, values beyond
// the half range become infinity, every NaN becomes the quiet NaN 0x7e00
uint16_t float_to_half(float value) {
	uint32_t bits; memcpy(&bits, &value, 4);
	uint32_t sign = (bits >> 16) & 0x8000u;
	uint32_t biased = (bits >> 23) & 0xff;
	uint32_t mantissa = bits & 0x7fffffu;

	if (biased == 0)    return uint16_t(sign);
	if (biased == 0xff) return uint16_t(sign | 0x7c00u | (mantissa ? 0x200u : 0u));
	int32_t exponent = int32_t(biased) - 127 + 15;
	if (exponent >= 31) return uint16_t(sign | 0x7c00u);
	if (exponent <= 0) {
		if (14 - exponent > 24) return uint16_t(sign); // nothing of the mantissa is left
		mantissa |= 0x800000u;                         // make the leading one explicit, then shift it into place
		uint32_t half = mantissa >> (14 - exponent);
		if ((mantissa >> (13 - exponent)) & 1u) half++;
		return uint16_t(sign | half);
	}
	uint32_t half = (uint32_t(exponent) << 10) | (mantissa >> 13);
	if (mantissa & 0x1000u) half++; // may carry into the exponent, up to infinity
	return uint16_t(sign | half);
}

struct Bytes {
	std::vector<unsigned char> data;
	void raw(const void * p, size_t n) { const unsigned char * b = (const unsigned char *)p; data.insert(data.end(), b, b + n); }
	void u8 (unsigned char v) { data.push_back(v); }
	void i32(int32_t v)  { raw(&v, 4); }
	void u64(uint64_t v) { raw(&v, 8); }
	void f32(float v)    { raw(&v, 4); }
	void str(const char * s) { raw(s, strlen(s) + 1); }
	void attribute(const char * name, const char * type, const Bytes & value) {
		str(name); str(type); i32(int32_t(value.data.size())); raw(value.data.data(), value.data.size());
	}
};
}

Vector3 Exporters::tonemap(Vector3 colour) {
	float channel[3] = { colour.x, colour.y, colour.z };
	for (float & c : channel) {
		c = c > 0.0f ? c : 0.0f;
		c = clamp((c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f), 0.0f, 1.0f);
		c = powf(c, 1.0f / 2.2f);
		c = floorf(c * 255.0f + 0.5f) / 255.0f; // stored in the 8-bit back buffer, read back as float
	}
	return Vector3(channel[0], channel[1], channel[2]);
}

bool PPMExporter::save(const std::string & filename, int pitch, int width, int height, const std::vector<Vector3> & data) {
	std::vector<unsigned char> bytes;
	bytes.reserve(size_t(width) * height * 3);
	for (int y = height - 1; y >= 0; y--) {
		for (int x = 0; x < width; x++) {
			const Vector3 & c = data[x + size_t(y) * pitch];
			bytes.push_back((unsigned char)(clamp(c.x * 255.0f, 0.0f, 255.0f)));
			bytes.push_back((unsigned char)(clamp(c.y * 255.0f, 0.0f, 255.0f)));
			bytes.push_back((unsigned char)(clamp(c.z * 255.0f, 0.0f, 255.0f)));
		}
	}
	File file(filename, "wb");
	if (!file.f) return false;
	fprintf(file.f, "P6\n %d\n %d\n %d\n", width, height, 255);
	return fwrite(bytes.data(), 1, bytes.size(), file.f) == bytes.size();
}

bool EXRExporter::save(const std::string & filename, int pitch, int width, int height, const std::vector<Vector3> & data) {
	Bytes out;
	const unsigned char magic_and_version[8] = { 0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0 };
	out.raw(magic_and_version, 8);

	Bytes channels;
	for (const char * name : { "B", "G", "R" }) { // alphabetical, as the format requires
		channels.str(name);
		channels.i32(1);            // pixel type HALF
		channels.u8(0); channels.u8(0); channels.u8(0); channels.u8(0); // pLinear + reserved
		channels.i32(1); channels.i32(1); // x / y sampling
	}
	channels.u8(0);
	out.attribute("channels", "chlist", channels);

	Bytes compression; compression.u8(0); // NO_COMPRESSION: one scan line per chunk
	out.attribute("compression", "compression", compression);
	Bytes window; window.i32(0); window.i32(0); window.i32(width - 1); window.i32(height - 1);
	out.attribute("dataWindow",    "box2i", window);
	out.attribute("displayWindow", "box2i", window);
	Bytes line_order; line_order.u8(0); // increasing y
	out.attribute("lineOrder", "lineOrder", line_order);
	Bytes one; one.f32(1.0f);
	out.attribute("pixelAspectRatio", "float", one);
	Bytes centre; centre.f32(0.0f); centre.f32(0.0f);
	out.attribute("screenWindowCenter", "v2f", centre);
	out.attribute("screenWindowWidth", "float", one);
	out.u8(0); // end of header

	size_t row_bytes   = size_t(width) * 3 * sizeof(uint16_t);
	size_t chunk_bytes = 8 + row_bytes;
	uint64_t first_chunk = out.data.size() + size_t(height) * 8;
	for (int row = 0; row < height; row++) out.u64(first_chunk + uint64_t(row) * chunk_bytes);

	std::vector<uint16_t> line(size_t(width) * 3);
	for (int row = 0; row < height; row++) {
		const Vector3 * src = data.data() + size_t(height - 1 - row) * pitch; // file row 0 is the top of the image
		for (int x = 0; x < width; x++) {
			line[x]             = float_to_half(src[x].z);
			line[x + width]     = float_to_half(src[x].y);
			line[x + 2 * width] = float_to_half(src[x].x);
		}
		out.i32(row);
		out.i32(int32_t(row_bytes));
		out.raw(line.data(), row_bytes);
	}
	File file(filename, "wb");
	if (!file.f) return false;
	return fwrite(out.data.data(), 1, out.data.size(), file.f) == out.data.size();
}

bool Exporters::save(const std::string & filename, int pitch, int width, int height, const std::vector<Vector3> & radiance, std::string * error) {
	size_t dot = filename.find_last_of('.');
	std::string extension = dot == std::string::npos ? std::string() : filename.substr(dot + 1);
	bool ok;
	if (extension == "ppm") {
		std::vector<Vector3> display(radiance.size());
		for (size_t i = 0; i < radiance.size(); i++) display[i] = tonemap(radiance[i]);
		ok = PPMExporter::save(filename, pitch, width, height, display);
	} else if (extension == "exr") {
		ok = EXRExporter::save(filename, pitch, width, height, radiance);
	} else {
		if (error) *error = "unsupported output file extension '" + extension + "' (ppm and exr are supported)";
		return false;
	}
	if (!ok && error) *error = "failed to write '" + filename + "'";
	return ok;
}
