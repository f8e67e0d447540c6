// Minimal text cursor used by the OBJ and Mitsuba-XML loaders.
// parse_float reproduces the reference's digit-accumulating double parser
// (Src/Core/Parser.h:187-252) instead of strtof so that vertex coordinates
// round to the same floats the reference would load.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <string_view>

struct ParseError : std::exception {
	std::string message;
	explicit ParseError(std::string m) : message(std::move(m)) { }
	const char * what() const noexcept override { return message.c_str(); }
};

inline bool is_digit(char c)      { return c >= '0' && c <= '9'; }
inline bool is_whitespace(char c) { return c == ' ' || c == '\t'; }
inline bool is_newline(char c)    { return c == '\r' || c == '\n'; }

struct Parser {
	const char * cur;
	const char * start;
	const char * end;
	std::string  filename;
	int line = 1;

	Parser(std::string_view data, std::string filename = "") : cur(data.data()), start(data.data()), end(data.data() + data.size()), filename(std::move(filename)) { }

	[[noreturn]] void fail(const std::string & what) const {
		throw ParseError(filename + ":" + std::to_string(line) + ": " + what);
	}

	bool reached_end() const { return cur >= end; }
	char peek() const { return cur < end ? *cur : '\0'; }

	void advance(int n = 1) {
		if (cur + n > end) fail("unexpected end of file");
		for (int i = 0; i < n; i++) { if (*cur == '\n') line++; cur++; }
	}
	void skip_whitespace()            { while (cur < end && is_whitespace(*cur)) advance(); }
	void skip_whitespace_or_newline() { while (cur < end && (is_whitespace(*cur) || is_newline(*cur))) advance(); }
	void skip_line()                  { while (cur < end && !is_newline(*cur)) advance(); }

	bool match(char c) { if (cur < end && *cur == c) { advance(); return true; } return false; }
	bool match(std::string_view s) {
		if (size_t(end - cur) >= s.size() && std::string_view(cur, s.size()) == s) { advance(int(s.size())); return true; }
		return false;
	}
	void expect(char c) {
		if (reached_end()) fail(std::string("unexpected end of file, expected '") + c + "'");
		if (*cur != c) fail(std::string("unexpected char '") + *cur + "', expected '" + c + "'");
		advance();
	}

	int parse_int() {
		bool negative = match('-');
		if (!negative) match('+');
		if (!is_digit(peek())) fail("expected integer digit");
		unsigned value = 0; // unsigned: an absurdly long digit string wraps around instead of overflowing
		while (is_digit(peek())) { value = value * 10u + unsigned(*cur - '0'); advance(); }
		return int(negative ? 0u - value : value);
	}

	float parse_float() {
		if (match("nan") || match("NAN")) return NAN;
		bool negative = false;
		if (match('-')) negative = true; else match('+');
		skip_whitespace();
		if (match("inf") || match("INF") || match("infinity") || match("INFINITY")) return negative ? -INFINITY : INFINITY;

		double value = 0.0;
		bool has_int = false, has_frac = false;
		if (is_digit(peek())) { value = parse_int(); has_int = true; }
		if (match('.')) {
			static constexpr double SCALE[] = { 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001, 0.00000001, 0.000000001, 0.0000000001, 0.00000000001 };
			int digit = 0;
			while (is_digit(peek())) {
				double p = digit < int(sizeof(SCALE) / sizeof(SCALE[0])) ? SCALE[digit] : pow(0.1, digit);
				value += double(*cur - '0') * p;
				digit++;
				advance();
			}
			has_frac = digit > 0;
		}
		if (!has_int && !has_frac) fail("expected float");
		if (match('e') || match('E')) {
			int exponent = parse_int();
			value = value * pow(10.0, exponent);
		}
		return float(negative ? -value : value);
	}
};
