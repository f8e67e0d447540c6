// Typed index into one of the AssetManager arrays (reference: Src/Renderer/Handle.h).
#pragma once
#include "Math.h"

template<typename T>
struct Handle {
	int handle = INVALID;
	static Handle<T> get_default() { return Handle<T> { 0 }; }
};
template<typename T> bool operator==(Handle<T> a, Handle<T> b) { return a.handle == b.handle; }
template<typename T> bool operator!=(Handle<T> a, Handle<T> b) { return a.handle != b.handle; }
