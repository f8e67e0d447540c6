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

// The document is read by ONE loop over its tags with an explicit stack of the elements that are still open (no recursion:
// a scene file nested a few thousand levels deep is a malformed file, not a stack overflow). A tag is one of
//   <name attr="v" ...>    opens an element: pushed      <name ... />  or  <?name ... ?>   a complete element
//   </name>                closes the innermost open element, whose name it must repeat
// Comments may stand wherever white space may. No entities, no CDATA, no text content (Mitsuba scene files have none).
namespace {

struct TagScanner {
	Parser & in;

	void blank() { parser_skip_xml_whitespace(in); }
	bool at_end() const { return in.reached_end(); }

	// characters up to (not including) the first one `stop` accepts
	template<typename Stop> std::string until(Stop stop) {
		const char * from = in.cur;
		while (!in.reached_end() && !stop(*in.cur)) in.advance();
		return std::string(from, in.cur);
	}

	XMLAttribute attribute() {
		XMLAttribute a;
		a.name = until([](char c) { return c == '='; });
		while (!a.name.empty() && (is_whitespace(a.name.back()) || is_newline(a.name.back()))) a.name.pop_back();
		in.expect('=');
		(void)until([](char c) { return c == '"' || c == '\''; });
		if (in.reached_end()) in.fail("an attribute value must be quoted");
		const char quote = *in.cur;
		in.advance();
		a.value = until([quote](char c) { return c == quote; });
		in.expect(quote);
		return a;
	}
};

} // namespace

XMLNode XMLParser::parse_root() {
	XMLNode root;
	root.location = parser.filename;
	TagScanner scan { parser };
	std::vector<XMLNode> open;   // the elements whose closing tag has not been seen yet, outermost first

	auto finished = [&](XMLNode && element) {
		(open.empty() ? root : open.back()).children.push_back(std::move(element));
	};

	while (true) {
		scan.blank();
		if (scan.at_end()) {
			if (!open.empty()) parser.fail("missing closing tag for <" + open.back().tag + ">");
			return root;
		}
		if (parser.match("</")) {   // closes the innermost element
			std::string name = scan.until([](char c) { return c == '>'; });
			parser.expect('>');
			if (open.empty()) parser.fail("closing tag '" + name + "' without an open element");
			if (name != open.back().tag) parser.fail("non matching closing tag '" + name + "' for node '" + open.back().tag + "'");
			XMLNode element = std::move(open.back());
			open.pop_back();
			finished(std::move(element));
			continue;
		}

		parser.expect('<');
		XMLNode element;
		element.location = parser.filename + ":" + std::to_string(parser.line);
		element.is_question_mark = parser.match('?');
		element.tag = scan.until([](char c) { return is_whitespace(c) || is_newline(c) || c == '>' || c == '/'; });
		if (element.tag.empty()) parser.fail("empty open tag");

		bool complete = false;      // <name ... /> or <?name ... ?>
		while (true) {
			scan.blank();
			if (scan.at_end()) parser.fail("unterminated tag <" + element.tag + ">");
			if (parser.match('>')) break;
			if (parser.match('/') || (element.is_question_mark && parser.match('?'))) { parser.expect('>'); complete = true; break; }
			element.attributes.push_back(scan.attribute());
		}
		if (complete) finished(std::move(element));
		else open.push_back(std::move(element));
	}
}
