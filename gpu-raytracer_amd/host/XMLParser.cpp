#include "XMLParser.h"

#include <fstream>
#include <sstream>

std::string read_text_file(const std::string & filename) {
	std::ifstream f(filename, std::ios::binary);
	if (!f) throw ParseError("unable to open '" + filename + "'");
	std::stringstream ss;
	ss << f.rdbuf();
	return ss.str();
}

void parser_skip_xml_whitespace(Parser & parser) {
	parser.skip_whitespace_or_newline();
	while (parser.match("<!--")) {
		while (!parser.reached_end() && !parser.match("-->")) parser.advance();
		parser.skip_whitespace_or_newline();
	}
}

bool XMLAttribute::as_bool() const {
	if (value == "true")  return true;
	if (value == "false") return false;
	throw ParseError("unable to parse '" + value + "' as boolean");
}

Vector3 XMLAttribute::as_vector3() const {
	Parser p(value);
	parser_skip_xml_whitespace(p);
	Vector3 v;
	v.x = p.parse_float();
	bool uses_comma = p.match(',');
	parser_skip_xml_whitespace(p);
	if (!p.reached_end()) {
		v.y = p.parse_float();
		if (uses_comma) p.expect(',');
		parser_skip_xml_whitespace(p);
		v.z = p.parse_float();
	} else {
		v.y = v.x; // a single number broadcasts
		v.z = v.x;
	}
	return v;
}

Matrix4 XMLAttribute::as_matrix4() const {
	Parser p(value);
	Matrix4 m;
	for (int i = 0; i < 16; i++) {
		parser_skip_xml_whitespace(p);
		m.cells[i] = p.parse_float();
	}
	return m;
}

XMLParser::XMLParser(const std::string & filename) : source(read_text_file(filename)), parser(source, filename) { }

XMLNode XMLParser::parse_root() {
	XMLNode root;
	root.location = parser.filename;
	while (!parser.reached_end()) {
		parser_skip_xml_whitespace(parser);
		if (parser.reached_end()) break;
		root.children.push_back(parse_tag());
		parser_skip_xml_whitespace(parser);
	}
	return root;
}

XMLNode XMLParser::parse_tag() {
	XMLNode node;
	if (parser.reached_end()) return node;

	parser.expect('<');
	node.location = parser.filename + ":" + std::to_string(parser.line);
	node.is_question_mark = parser.match('?');

	const char * tag_start = parser.cur;
	while (!parser.reached_end() && !is_whitespace(*parser.cur) && !is_newline(*parser.cur) && *parser.cur != '>' && *parser.cur != '/') parser.advance();
	node.tag.assign(tag_start, parser.cur);
	if (node.tag.empty()) parser.fail("empty open tag");

	parser_skip_xml_whitespace(parser);

	// Attributes, until '>' (children follow) or '/>' / '?>' (inline tag)
	while (true) {
		if (parser.reached_end()) parser.fail("unterminated tag <" + node.tag + ">");
		if (parser.match('>')) break;
		if (parser.match('/') || (node.is_question_mark && parser.match('?'))) { parser.expect('>'); return node; }

		XMLAttribute attribute;
		const char * name_start = parser.cur;
		while (!parser.reached_end() && *parser.cur != '=') parser.advance();
		attribute.name.assign(name_start, parser.cur);
		while (!attribute.name.empty() && (is_whitespace(attribute.name.back()) || is_newline(attribute.name.back()))) attribute.name.pop_back();

		parser.expect('=');
		while (!parser.reached_end() && *parser.cur != '"' && *parser.cur != '\'') parser.advance();
		char quote;
		if      (parser.match('"'))  quote = '"';
		else if (parser.match('\'')) quote = '\'';
		else parser.fail("an attribute value must be quoted");

		const char * value_start = parser.cur;
		while (!parser.reached_end() && *parser.cur != quote) parser.advance();
		attribute.value.assign(value_start, parser.cur);
		parser.expect(quote);
		parser_skip_xml_whitespace(parser);

		node.attributes.push_back(std::move(attribute));
	}

	parser_skip_xml_whitespace(parser);
	while (!parser.match("</")) {
		if (parser.reached_end()) parser.fail("missing closing tag for <" + node.tag + ">");
		node.children.push_back(parse_tag());
		parser_skip_xml_whitespace(parser);
	}

	const char * closing_start = parser.cur;
	while (!parser.reached_end() && *parser.cur != '>') parser.advance();
	std::string closing(closing_start, parser.cur);
	parser.expect('>');
	if (closing != node.tag) parser.fail("non matching closing tag '" + closing + "' for node '" + node.tag + "'");
	return node;
}
