// Sample tables for random<Dim>() (reference: Src/Util/PMJ.h, Src/Util/BlueNoise.h).
#pragma once
#include <string>
#include <vector>

namespace PMJ {
	// 64 sequences x 4096 points, x/y interleaved floats in [0,1).
	// The reference's table (Src/Util/PMJ.cpp) is not part of the mount
	// (.MISSING_LARGE_BLOBS:13), so its contents are regenerated: each sequence is an
	// Owen-scrambled Sobol' (0,2)-sequence, which has the pmj02 property that every
	// power-of-two prefix is stratified over all elementary intervals (Christensen et
	// al. 2018, section 5).  Deterministic: depends only on `seed`.
	std::vector<float> generate(unsigned seed = 0x9e3779b9u);
}

namespace BlueNoise {
	// 16 tiles x 128 x 128 x {x,y} bytes; read from assets/blue_noise_16x128x128_rg8.bin
	// (searched relative to the library, $GRT_ASSET_DIR, then the working directory).
	std::vector<unsigned char> load();
	std::string asset_directory();
}
