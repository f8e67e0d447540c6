// Small DOM-style XML reader for Mitsuba 0.5 scene files: tags, quoted attributes,
// comments, self-closing tags; no entities/CDATA (reference: Src/Assets/Mitsuba/XMLParser.{h,cpp}).
#pragma once
#include <string>
#include <string_view>
#include <vector>

#include "Parser.h"
#include "Math.h"

void parser_skip_xml_whitespace(Parser & parser);

struct XMLAttribute {
	std::string name;
	std::string value;

	int     as_int()   const { Parser p(value); return p.parse_int(); }
	float   as_float() const { Parser p(value); return p.parse_float(); }
	bool    as_bool()  const;
	Vector3 as_vector3() const; // "x", "x y z" or "x, y, z"
	Matrix4 as_matrix4() const; // 16 row-major floats
};

struct XMLNode {
	std::string tag;
	bool is_question_mark = false;
	std::vector<XMLAttribute> attributes;
	std::vector<XMLNode>      children;
	std::string location;

	const XMLAttribute * get_attribute(std::string_view name) const {
		for (const XMLAttribute & a : attributes) if (a.name == name) return &a;
		return nullptr;
	}
	const XMLAttribute & require_attribute(std::string_view name) const {
		if (const XMLAttribute * a = get_attribute(name)) return *a;
		throw ParseError(location + ": node '" + tag + "' does not have an attribute with name '" + std::string(name) + "'");
	}
	std::string_view get_attribute_value(std::string_view name) const { return require_attribute(name).value; }

	float   get_attribute_optional(std::string_view name, float   def) const { const XMLAttribute * a = get_attribute(name); return a ? a->as_float()   : def; }
	Vector3 get_attribute_optional(std::string_view name, Vector3 def) const { const XMLAttribute * a = get_attribute(name); return a ? a->as_vector3() : def; }
	int     get_attribute_optional(std::string_view name, int     def) const { const XMLAttribute * a = get_attribute(name); return a ? a->as_int()     : def; }

	const XMLNode * get_child_by_tag(std::string_view t) const {
		for (const XMLNode & c : children) if (c.tag == t) return &c;
		return nullptr;
	}
	// First child whose name="..." attribute equals 'name'
	const XMLNode * get_child_by_name(std::string_view name) const {
		for (const XMLNode & c : children) {
			const XMLAttribute * a = c.get_attribute("name");
			if (a && a->value == name) return &c;
		}
		return nullptr;
	}
	const XMLNode & require_child_by_name(std::string_view name) const {
		if (const XMLNode * c = get_child_by_name(name)) return *c;
		throw ParseError(location + ": node '" + tag + "' does not have a child with name '" + std::string(name) + "'");
	}

	float   get_child_value_optional(std::string_view name, float   def) const { const XMLNode * c = get_child_by_name(name); return c ? c->get_attribute_optional("value", def) : def; }
	Vector3 get_child_value_optional(std::string_view name, Vector3 def) const { const XMLNode * c = get_child_by_name(name); return c ? c->get_attribute_optional("value", def) : def; }
	int     get_child_value_optional(std::string_view name, int     def) const { const XMLNode * c = get_child_by_name(name); return c ? c->get_attribute_optional("value", def) : def; }
};

struct XMLParser {
	std::string source;
	Parser parser;

	explicit XMLParser(const std::string & filename);
	XMLNode parse_root();   // the whole document: a nameless node whose children are the top-level elements
};

std::string read_text_file(const std::string & filename);
