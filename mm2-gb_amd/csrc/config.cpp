// config.cpp -- gpu_config.json reader.
// The reference parses this file with cJSON (an un-vendored submodule; plmem.cu:372-405) and pulls out the keys listed
// at plmem.cu:416-443 and :486-495.  This is a small self-contained JSON reader that accepts the same documents: the five
// shipped presets (gpu/*.json) load unchanged; keys starting with "//" are comments; unknown keys are ignored.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "engine.h"

namespace mm2gb {
namespace {

struct JVal {
	enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
	double num = 0; bool b = false; std::string str;
	std::vector<JVal> arr;
	std::vector<std::pair<std::string, JVal>> obj;   // keeps duplicates; first match wins like cJSON_GetObjectItem
	const JVal *get(const char *key) const
	{
		for (auto &kv : obj) if (kv.first == key) return &kv.second;
		return nullptr;
	}
};

struct Parser {
	const char *s; size_t pos = 0, len; std::string err;
	explicit Parser(const char *text) : s(text), len(strlen(text)) {}
	void ws() { while (pos < len && isspace((unsigned char)s[pos])) ++pos; }
	bool fail_at(const std::string &what)
	{
		if (err.empty()) err = what + " at byte " + std::to_string(pos);
		return false;
	}
	bool parse_string(std::string &out)
	{
		if (s[pos] != '"') return fail_at("expected string");
		++pos; out.clear();
		while (pos < len && s[pos] != '"') {
			char c = s[pos++];
			if (c == '\\') {
				if (pos >= len) return fail_at("bad escape");
				char e = s[pos++];
				switch (e) {
				case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
				case 'b': out += '\b'; break; case 'f': out += '\f'; break;
				case 'u': { if (pos + 4 > len) return fail_at("bad \\u escape"); out += '?'; pos += 4; break; }
				default: out += e;
				}
			} else out += c;
		}
		if (pos >= len) return fail_at("unterminated string");
		++pos;
		return true;
	}
	bool parse_value(JVal &v, int depth)
	{
		if (depth > 64) return fail_at("nesting too deep");
		ws();
		if (pos >= len) return fail_at("unexpected end of input");
		char c = s[pos];
		if (c == '{') {
			v.kind = JVal::Obj; ++pos; ws();
			if (pos < len && s[pos] == '}') { ++pos; return true; }
			while (true) {
				ws();
				std::string key;
				if (!parse_string(key)) return false;
				ws();
				if (pos >= len || s[pos] != ':') return fail_at("expected ':'");
				++pos;
				JVal child;
				if (!parse_value(child, depth + 1)) return false;
				v.obj.emplace_back(std::move(key), std::move(child));
				ws();
				if (pos < len && s[pos] == ',') { ++pos; continue; }
				if (pos < len && s[pos] == '}') { ++pos; return true; }
				return fail_at("expected ',' or '}'");
			}
		}
		if (c == '[') {
			v.kind = JVal::Arr; ++pos; ws();
			if (pos < len && s[pos] == ']') { ++pos; return true; }
			while (true) {
				JVal child;
				if (!parse_value(child, depth + 1)) return false;
				v.arr.push_back(std::move(child));
				ws();
				if (pos < len && s[pos] == ',') { ++pos; continue; }
				if (pos < len && s[pos] == ']') { ++pos; return true; }
				return fail_at("expected ',' or ']'");
			}
		}
		if (c == '"') { v.kind = JVal::Str; return parse_string(v.str); }
		if (!strncmp(s + pos, "true", 4)) { v.kind = JVal::Bool; v.b = true; pos += 4; return true; }
		if (!strncmp(s + pos, "false", 5)) { v.kind = JVal::Bool; v.b = false; pos += 5; return true; }
		if (!strncmp(s + pos, "null", 4)) { v.kind = JVal::Null; pos += 4; return true; }
		if (c == '-' || isdigit((unsigned char)c)) {
			char *end = nullptr;
			v.num = strtod(s + pos, &end);
			if (end == s + pos) return fail_at("bad number");
			v.kind = JVal::Num; pos = (size_t)(end - s);
			return true;
		}
		return fail_at(std::string("unexpected character '") + c + "'");
	}
};

// cJSON's valueint saturates at INT_MAX/INT_MIN (the reference reads every field but max_total_n through it)
int to_int_sat(double d)
{
	if (d >= 2147483647.0) return 2147483647;
	if (d <= -2147483648.0) return (-2147483647 - 1);
	return (int)d;
}

bool want_int(const JVal *obj, const char *key, int *out, bool required, std::string &err)
{
	const JVal *v = obj ? obj->get(key) : nullptr;
	if (!v) {
		if (required) { err = std::string("gpu config: failed to get field ") + key; return false; }
		return true;
	}
	if (v->kind != JVal::Num) { err = std::string("gpu config: field ") + key + " is not a number"; return false; }
	*out = to_int_sat(v->num);
	return true;
}

} // namespace
} // namespace mm2gb

using namespace mm2gb;

extern "C" {

void mm2gb_config_defaults(mm2gb_config_t *c)
{
	memset(c, 0, sizeof(*c));
	// sized for one MI355X (288 GB HBM3E, 256 CUs): see mm2-gb_amd/mi355x_config.json for the same numbers as a file
	c->num_streams = 1;
	c->min_n = 512;
	c->long_seg_buffer_size = 1000000000;
	c->max_total_n = 500000000;
	c->max_read = 2000000;
	c->avg_read_n = 20000;
	c->has_max_total_n = c->has_max_read = 1; c->has_avg_read_n = 0;
	c->range_kernel.blockdim = 256;
	c->range_kernel.cut_check_anchors = 10;
	c->range_kernel.anchor_per_block = 1024;
	c->score_kernel.micro_batch = 4;
	c->score_kernel.mid_blockdim = 512;
	c->score_kernel.short_griddim = 2048;
	c->score_kernel.mid_griddim = 2048;
	c->score_kernel.long_griddim = 256;
	c->score_kernel.long_seg_cutoff = 20;
	c->score_kernel.mid_seg_cutoff = 3;
}

int mm2gb_config_parse(const char *json_text, mm2gb_config_t *c)
{
	if (!json_text || !c) return fail("mm2gb_config_parse: null argument");
	mm2gb_config_defaults(c);
	Parser ps(json_text);
	JVal root;
	if (!ps.parse_value(root, 0)) return fail("gpu config: JSON error: " + ps.err);
	ps.ws();
	if (ps.pos != ps.len) return fail("gpu config: trailing characters after JSON document at byte " + std::to_string(ps.pos));
	if (root.kind != JVal::Obj) return fail("gpu config: top level must be an object");
	std::string err;
	// top level (plmem.cu:473-495): num_streams, min_n required; max_total_n/max_read/long_seg_buffer_size or avg_read_n
	if (!want_int(&root, "num_streams", &c->num_streams, true, err)) return fail(err);
	if (!want_int(&root, "min_n", &c->min_n, true, err)) return fail(err);
	const JVal *mt = root.get("max_total_n"), *mr = root.get("max_read"), *lb = root.get("long_seg_buffer_size"), *av = root.get("avg_read_n");
	c->has_max_total_n = mt && mt->kind == JVal::Num;
	c->has_max_read = mr && mr->kind == JVal::Num;
	c->has_avg_read_n = av && av->kind == JVal::Num;
	if (c->has_max_total_n) c->max_total_n = (int64_t)mt->num;             // valuedouble, plmem.cu:491
	if (c->has_max_read) c->max_read = to_int_sat(mr->num);
	if (lb && lb->kind == JVal::Num) c->long_seg_buffer_size = to_int_sat(lb->num);   // valueint, plmem.cu:493
	if (c->has_avg_read_n) c->avg_read_n = to_int_sat(av->num);
	if (!(c->has_max_total_n && c->has_max_read) && !c->has_avg_read_n)
		return fail("gpu config: need max_total_n and max_read, or avg_read_n");
	// kernels (plmem.cu:416-443): all fields required there
	const JVal *rk = root.get("range_kernel"), *sk = root.get("score_kernel");
	if (!rk || rk->kind != JVal::Obj) return fail("gpu config: failed to get field range_kernel");
	if (!sk || sk->kind != JVal::Obj) return fail("gpu config: failed to get field score_kernel");
	if (!want_int(rk, "blockdim", &c->range_kernel.blockdim, true, err) ||
	    !want_int(rk, "cut_check_anchors", &c->range_kernel.cut_check_anchors, true, err) ||
	    !want_int(rk, "anchor_per_block", &c->range_kernel.anchor_per_block, true, err) ||
	    !want_int(sk, "mid_blockdim", &c->score_kernel.mid_blockdim, true, err) ||
	    !want_int(sk, "short_griddim", &c->score_kernel.short_griddim, true, err) ||
	    !want_int(sk, "long_griddim", &c->score_kernel.long_griddim, true, err) ||
	    !want_int(sk, "mid_griddim", &c->score_kernel.mid_griddim, true, err) ||
	    !want_int(sk, "long_seg_cutoff", &c->score_kernel.long_seg_cutoff, true, err) ||
	    !want_int(sk, "mid_seg_cutoff", &c->score_kernel.mid_seg_cutoff, true, err) ||
	    !want_int(sk, "micro_batch", &c->score_kernel.micro_batch, true, err))
		return fail(err);
	if (c->score_kernel.micro_batch < 1) return fail("gpu config: score_kernel:micro_batch must be >= 1");
	if (c->num_streams < 1) return fail("gpu config: num_streams must be >= 1");
	return 0;
}

int mm2gb_config_load(const char *path, mm2gb_config_t *c)
{
	if (!path || !c) return fail("mm2gb_config_load: null argument");
	FILE *fp = fopen(path, "rb");
	if (!fp) return fail(std::string("fail to open gpu config file ") + path);   // message of plmem.cu:390-393
	std::string text;
	char buf[4096];
	size_t got;
	while ((got = fread(buf, 1, sizeof(buf), fp)) > 0) text.append(buf, got);
	fclose(fp);
	return mm2gb_config_parse(text.c_str(), c);
}

} // extern "C"
