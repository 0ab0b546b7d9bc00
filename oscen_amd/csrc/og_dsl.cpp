// og_dsl.cpp -- text front end for the body of a reference `graph! { ... }` invocation.
//
// Grammar followed (oscen-graph-compiler/src/parse.rs:195-979):
//   name: Ident;                                   nih_params;   (ignored)
//   input  NAME: KIND [: Type] [= default] [[spec] | {spec}];     (old form: input KIND NAME ...)
//   output NAME: KIND [: Type];                                    (old form: output KIND NAME;)
//   nodes { NAME = path::Type::ctor(args) [* N]; ... }   also `node { }` and `node NAME = ...;`
//   connections { [policy] SRC_EXPR -> DST; ... }        also `connection { }` / `connection ...;`
// KIND = value | event | stream; the only part of a [spec] the engine needs is `ramp: N`;
// policy = latch | linear | sinc | sinc_iir; DST = node.port | node.port() | output name.
//   nodes { NAME = [path::Type::ctor(args); N] [* M]; }  node arrays inside the voice graph (parse.rs:447-520)
//   nodes { NAME = GraphType; }  /  GraphType::new()      a registered graph type used as a node (nested graph)
// Not handled (diagnosed): `external`.  `src -> [N] -> dst` and `src -> [delay_node] -> dst` expand into the two edges of
// ir/lower.rs:342-347 (the second one a feedback edge).
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>
#include <sstream>
#include <stdexcept>

#include "og_graph.h"

namespace ogc {
namespace {

[[noreturn]] void dfail_unsupported(const std::string& m, size_t line) { throw ogabi::Unsupported("oscen graph dsl: line " + std::to_string(line) + ": " + m); }
[[noreturn]] void dfail(const std::string& m, size_t line) { throw std::runtime_error("oscen graph dsl: line " + std::to_string(line) + ": " + m); }

struct Lexer {
    std::string s;
    size_t i = 0, line = 1;
    explicit Lexer(std::string t) : s(std::move(t)) {}
    void skip()
    {
        for (;;) {
            while (i < s.size() && isspace((unsigned char)s[i])) {
                if (s[i] == '\n') ++line;
                ++i;
            }
            if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') {
                while (i < s.size() && s[i] != '\n') ++i;
                continue;
            }
            if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '*') {
                i += 2;
                while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) {
                    if (s[i] == '\n') ++line;
                    ++i;
                }
                i += 2;
                continue;
            }
            return;
        }
    }
    bool eof()
    {
        skip();
        return i >= s.size();
    }
    char peek()
    {
        skip();
        return i < s.size() ? s[i] : '\0';
    }
    bool eat(char c)
    {
        if (peek() == c) {
            ++i;
            return true;
        }
        return false;
    }
    void expect(char c)
    {
        if (!eat(c)) dfail(std::string("expected '") + c + "'", line);
    }
    bool eat_arrow()
    {
        skip();
        if (i + 1 < s.size() && s[i] == '-' && s[i + 1] == '>') {
            i += 2;
            return true;
        }
        return false;
    }
    std::string ident()
    {
        skip();
        size_t b = i;
        while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_')) ++i;
        if (b == i) dfail("expected identifier", line);
        return s.substr(b, i - b);
    }
    bool peek_ident(const char* kw)
    {
        skip();
        size_t n = strlen(kw);
        if (s.compare(i, n, kw) != 0) return false;
        return i + n >= s.size() || !(isalnum((unsigned char)s[i + n]) || s[i + n] == '_');
    }
    // raw text up to (not including) one of the stop characters at nesting depth 0
    std::string until(const char* stops)
    {
        skip();
        size_t b = i;
        int depth = 0;
        while (i < s.size()) {
            const char c = s[i];
            if (depth == 0 && strchr(stops, c)) break;
            if (c == '(' || c == '[' || c == '{' || c == '<') ++depth;
            if (c == ')' || c == ']' || c == '}' || c == '>') {
                if (depth == 0) break;
                --depth;
            }
            if (c == '\n') ++line;
            ++i;
        }
        std::string r = s.substr(b, i - b);
        while (!r.empty() && isspace((unsigned char)r.back())) r.pop_back();
        return r;
    }
    // raw text up to the next `->` (or `;`) at nesting depth 0
    std::string until_arrow()
    {
        skip();
        size_t b = i;
        int depth = 0;
        while (i < s.size()) {
            const char c = s[i];
            if (depth == 0 && (c == ';' || (c == '-' && i + 1 < s.size() && s[i + 1] == '>'))) break;
            if (c == '(' || c == '[') ++depth;
            if (c == ')' || c == ']') --depth;
            if (c == '\n') ++line;
            ++i;
        }
        std::string r = s.substr(b, i - b);
        while (!r.empty() && isspace((unsigned char)r.back())) r.pop_back();
        return r;
    }
    float number()
    {
        skip();
        std::string num;
        if (i < s.size() && (s[i] == '-' || s[i] == '+')) num.push_back(s[i++]);
        while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '.' || s[i] == '_' ||
                                ((s[i] == '-' || s[i] == '+') && !num.empty() && (num.back() == 'e' || num.back() == 'E')))) {
            if (s[i] != '_') num.push_back(s[i]);
            ++i;
        }
        for (const char* suf : {"f32", "f64"}) {
            size_t p = num.rfind(suf);
            if (p != std::string::npos && p + 3 == num.size()) num.erase(p);
        }
        char* end = nullptr;
        float v = strtof(num.c_str(), &end);
        if (num.empty() || end == num.c_str()) dfail("expected a number", line);
        return v;
    }
};

Kind kind_of(const std::string& k, size_t line)
{
    if (k == "value") return Kind::Value;
    if (k == "event") return Kind::Event;
    if (k == "stream") return Kind::Stream;
    dfail("unknown endpoint kind '" + k + "' (value | event | stream)", line);
}

uint32_t ramp_from_spec(const std::string& spec)
{ // `[0.0..2.0, ramp: 2205, unit = "s"]`
    size_t p = spec.find("ramp");
    if (p == std::string::npos) return 0;
    p = spec.find_first_of(":=", p);
    if (p == std::string::npos) return 0;
    std::string digits;
    for (++p; p < spec.size() && (isdigit((unsigned char)spec[p]) || spec[p] == '_' || spec[p] == ' '); ++p)
        if (isdigit((unsigned char)spec[p])) digits.push_back(spec[p]);
    return (uint32_t)strtoul(digits.c_str(), nullptr, 10);
}

void parse_input(Lexer& lx, GraphDesc& g)
{
    lx.ident(); // input
    std::string first = lx.ident(), name, kind;
    if (lx.eat(':')) {
        name = first;
        kind = lx.ident();
    } else {
        kind = first;
        name = lx.ident();
    }
    GInput in;
    in.name = name;
    in.kind = kind_of(kind, lx.line);
    if (lx.eat(':')) { // type annotation; `input stream dry: Frame<2>;` makes a stream input N channels wide
        std::string ty = lx.until("=;");
        ty.erase(std::remove_if(ty.begin(), ty.end(), [](char ch) { return isspace((unsigned char)ch); }), ty.end());
        if (in.kind == Kind::Stream) {
            if (ty == "Stereo") in.channels = 2;
            else if (ty == "Quad") in.channels = 4;
            else if (ty.rfind("Frame<", 0) == 0) in.channels = atoi(ty.c_str() + 6);
            if (in.channels < 1 || in.channels > 4) dfail("input '" + name + "': a stream input is an f32 or a Frame<2..4>", lx.line);
        }
    }
    if (lx.eat('=')) {
        in.def = lx.number();
        if (lx.peek() == '[') {
            lx.expect('[');
            in.ramp_frames = ramp_from_spec(lx.until("]"));
            lx.expect(']');
        } else if (lx.peek() == '{') {
            lx.expect('{');
            in.ramp_frames = ramp_from_spec(lx.until("}"));
            lx.expect('}');
        }
    }
    lx.expect(';');
    g.inputs.push_back(in);
}

void parse_output(Lexer& lx, GraphDesc& g)
{
    lx.ident(); // output
    std::string first = lx.ident(), name, kind;
    if (lx.eat(':')) {
        name = first;
        kind = lx.ident();
    } else {
        kind = first;
        name = lx.ident();
    }
    GOutput o{name, kind_of(kind, lx.line)};
    if (lx.eat(':')) { // `output out: stream: Frame<2>;` (examples/electric-piano/src/main.rs:51)
        std::string ty = lx.until(";");
        ty.erase(std::remove_if(ty.begin(), ty.end(), [](char ch) { return isspace((unsigned char)ch); }), ty.end());
        if (ty == "Stereo") o.channels = 2;
        else if (ty == "Quad") o.channels = 4;
        else if (ty == "f32" || ty == "Mono") o.channels = 1;
        else if (ty.rfind("Frame<", 0) == 0) o.channels = atoi(ty.c_str() + 6);
        else if (ty.rfind("[f32;", 0) == 0 && ty.back() == ']') o.channels = atoi(ty.c_str() + 5); // `[f32; N]` = N channels
        else dfail("output '" + name + "': unknown stream type '" + ty + "'", lx.line);
    }
    lx.expect(';');
    g.outputs.push_back(o);
}

void parse_node_decl(Lexer& lx, GraphDesc& g)
{
    GNode n;
    n.name = lx.ident();
    lx.expect('=');
    const bool is_array = lx.eat('[');
    // path::to::Type::ctor  -> keep the last two segments
    std::vector<std::string> segs;
    std::string generic;
    for (;;) {
        segs.push_back(lx.ident());
        lx.skip();
        if (lx.i + 1 < lx.s.size() && lx.s[lx.i] == ':' && lx.s[lx.i + 1] == ':') {
            lx.i += 2;
            if (lx.peek() == '<') { // turbofish `::<f32>` / `::<Frame<2>>` / `::<Stereo>`: the frame type of the node
                int depth = 0;
                generic.clear();
                do {
                    if (lx.i >= lx.s.size()) dfail("unterminated `::<...>`", lx.line);
                    const char ch = lx.s[lx.i++];
                    depth += ch == '<' ? 1 : (ch == '>' ? -1 : 0);
                    generic += ch;
                } while (depth > 0);
                generic = generic.substr(1, generic.size() - 2);
                lx.skip();
                if (lx.i + 1 < lx.s.size() && lx.s[lx.i] == ':' && lx.s[lx.i + 1] == ':') lx.i += 2;
                else break; // `VoiceAllocator::<4>;`: a bare generic type (= Type::<4>::new())
            }
            continue;
        }
        break;
    }
    if (lx.peek() == '(') {
        if (segs.size() < 2) dfail("expected `Type::ctor(...)` for node '" + n.name + "'", lx.line);
        n.type = segs[segs.size() - 2] + "::" + segs.back();
        if (!generic.empty()) n.type = normalize_type(segs[segs.size() - 2] + "::<" + generic + ">::" + segs.back());
        lx.expect('(');
        if (lx.peek() != ')') {
            for (;;) {
                // one argument: a number; `Frame([a, b, ..])` (a frame literal: its channels, in order); or any other
                // Rust expression, kept as text
                std::string raw = lx.until(",");
                Lexer al(raw);
                std::vector<float> vals;
                bool numeric = false;
                try {
                    if (al.peek_ident("Frame")) {
                        al.ident();
                        al.expect('(');
                        al.expect('[');
                        for (;;) {
                            vals.push_back(al.number());
                            if (al.eat(';')) { // `[x; N]`
                                const float rep = al.number();
                                const float x = vals.back();
                                vals.pop_back();
                                for (int k = 0; k < (int)rep; ++k) vals.push_back(x);
                                break;
                            }
                            if (!al.eat(',')) break;
                        }
                        al.expect(']');
                        al.expect(')');
                        numeric = al.eof();
                    } else {
                        vals.push_back(al.number());
                        numeric = al.eof();
                    }
                } catch (const std::exception&) {
                    numeric = false;
                }
                if (numeric) {
                    for (float v : vals) {
                        n.args.push_back(v);
                        if (!n.raw_args.empty()) n.raw_args.push_back("");
                    }
                } else {
                    if (raw.empty()) dfail("empty constructor argument", lx.line);
                    if (n.raw_args.empty()) n.raw_args.assign(n.args.size(), "");
                    n.args.push_back(0.0f);
                    n.raw_args.push_back(raw);
                }
                if (!lx.eat(',')) break;
            }
        }
        lx.expect(')');
    } else { // bare `Type` (a unit struct or a nested graph type: `inner = InnerGraph;`) = Type::new()
        n.type = segs.back() + "::new";
        if (!generic.empty()) n.type = normalize_type(segs.back() + "::<" + generic + ">::new");
    }
    if (is_array) {
        lx.expect(';');
        const float len = lx.number();
        if (!(len >= 1.0f) || len != (float)(uint32_t)len) dfail("node array length must be a positive integer", lx.line);
        n.array_len = (uint32_t)len;
        lx.expect(']');
    }
    if (lx.eat('*')) {
        n.rate_factor = (uint32_t)lx.number();
        // `[X; N] * M * P` (parse.rs:547-565; a second factor on a plain node is a type error in the reference's Rust)
        if (lx.peek() == '*' || lx.peek() == '/')
            dfail("node already has an embedded rate (`* N` or `/ N`)" + std::string(is_array ? " from the array literal" : "") +
                  "; remove the trailing rate annotation", lx.line);
    } else if (lx.peek() == '/') {
        dfail_unsupported("node undersampling (`/ N`) is not supported (neither is it by the reference v1)", lx.line);
    }
    lx.expect(';');
    g.nodes.push_back(n);
}

void parse_connection(Lexer& lx, GraphDesc& g)
{
    GEdge e;
    if (lx.peek() == '[') {
        lx.expect('[');
        e.policy = lx.ident();
        lx.expect(']');
    }
    e.src = lx.until_arrow();
    if (!lx.eat_arrow()) dfail("expected `->` in connection", lx.line);
    std::string via;
    if (lx.peek() == '[') { // `src -> [N] -> dst` / `src -> [delay_node] -> dst`  (ir/lower.rs:342-347)
        lx.expect('[');
        if (isdigit((unsigned char)lx.peek())) {
            const float n = lx.number();
            int k = 0;
            for (const auto& nd : g.nodes)
                if (nd.name.rfind("__inline_delay_", 0) == 0) ++k;
            GNode d;
            d.name = "__inline_delay_" + std::to_string(k);
            d.type = "Delay::new";
            d.args = {n, 0.0f};
            g.nodes.push_back(d);
            via = d.name;
        } else {
            via = lx.ident();
        }
        lx.expect(']');
        if (!lx.eat_arrow()) dfail("expected `->` after the delay in `src -> [..] -> dst`", lx.line);
    }
    e.dst = lx.until(";");
    size_t par = e.dst.find("()");
    if (par != std::string::npos) e.dst.erase(par, 2);
    lx.expect(';');
    if (e.src.empty() || e.dst.empty()) dfail("empty connection endpoint", lx.line);
    if (!via.empty()) {
        GEdge in_leg = e, out_leg = e;
        in_leg.dst = via + ".input";
        out_leg.src = via + ".output";
        out_leg.policy.clear();
        out_leg.feedback = true;
        g.edges.push_back(in_leg);
        g.edges.push_back(out_leg);
        return;
    }
    g.edges.push_back(e);
}

} // namespace

GraphDesc parse_dsl(const std::string& text, const std::vector<std::string>& per_voice)
{
    Lexer lx(text);
    GraphDesc g;
    g.name = "graph";
    // tolerate a full `graph! { ... }` wrapper
    if (lx.peek_ident("graph")) {
        size_t save = lx.i;
        lx.ident();
        if (lx.eat('!')) {
            lx.expect('{');
            size_t close = lx.s.rfind('}');
            if (close == std::string::npos) dfail("unterminated graph! { ... }", lx.line);
            lx.s.erase(close);
        } else {
            lx.i = save;
        }
    }
    while (!lx.eof()) {
        if (lx.peek_ident("name")) {
            lx.ident();
            lx.expect(':');
            g.name = lx.ident();
            lx.expect(';');
        } else if (lx.peek_ident("nih_params")) {
            lx.ident();
            lx.expect(';');
        } else if (lx.peek_ident("input")) {
            parse_input(lx, g);
        } else if (lx.peek_ident("output")) {
            parse_output(lx, g);
        } else if (lx.peek_ident("nodes") || lx.peek_ident("node")) {
            lx.ident();
            if (lx.eat('{')) {
                while (lx.peek() != '}') {
                    if (lx.eof()) dfail("unterminated nodes block", lx.line);
                    parse_node_decl(lx, g);
                }
                lx.expect('}');
            } else {
                parse_node_decl(lx, g);
            }
        } else if (lx.peek_ident("connections") || lx.peek_ident("connection")) {
            lx.ident();
            if (lx.eat('{')) {
                while (lx.peek() != '}') {
                    if (lx.eof()) dfail("unterminated connections block", lx.line);
                    parse_connection(lx, g);
                }
                lx.expect('}');
            } else {
                parse_connection(lx, g);
            }
        } else if (lx.peek_ident("external")) {
            dfail_unsupported("`external` asset handles are not supported", lx.line);
        } else {
            dfail("unexpected token '" + std::string(1, lx.peek()) + "'", lx.line);
        }
    }
    std::set<std::string> pv(per_voice.begin(), per_voice.end());
    for (auto& in : g.inputs)
        if (pv.count(in.name)) {
            if (in.kind != Kind::Value) throw std::runtime_error("oscen graph dsl: per-voice input '" + in.name + "' must be a value input");
            in.per_voice = true;
            pv.erase(in.name);
        }
    if (!pv.empty()) throw std::runtime_error("oscen graph dsl: per-voice input '" + *pv.begin() + "' is not declared");
    return g;
}

// Builder description -> DSL text (the voice-graph part; per-voice flags and bus nodes are noted in comments)
std::string to_dsl(const GraphDesc& g)
{
    std::ostringstream o;
    auto num = [](float v) {
        char b[64];
        snprintf(b, sizeof b, "%.9g", (double)v);
        std::string s = b;
        if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
        return s;
    };
    const char* kn[3] = {"value", "event", "stream"};
    o << "name: " << g.name << ";\n\n";
    for (const auto& in : g.inputs) {
        o << "input " << in.name << ": " << kn[(int)in.kind];
        if (in.kind == Kind::Stream && in.channels > 1) o << ": Frame<" << in.channels << ">";
        if (in.kind == Kind::Value) {
            o << " = " << num(in.def);
            if (in.ramp_frames) o << " [ramp: " << in.ramp_frames << "]";
        }
        o << ";" << (in.per_voice ? "  // per voice" : "") << "\n";
    }
    for (const auto& out : g.outputs)
        o << "output " << out.name << ": " << kn[(int)out.kind] << (out.channels > 1 ? ": Frame<" + std::to_string(out.channels) + ">" : std::string())
          << ";\n";
    o << "\nnodes {\n";
    for (const auto& n : g.nodes) {
        if (n.bus) {
            o << "    // post-mix (bus) node: " << n.name << " = " << n.type << "()\n";
            continue;
        }
        if (n.name.rfind("__inline_delay_", 0) == 0) continue; // printed as `-> [N] ->`
        std::string ty = normalize_type(n.type); // "TptFilter<2>::new" prints as the Rust path TptFilter::<Frame<2>>::new
        const size_t lt = ty.find('<'), gt = ty.find(">::");
        if (lt != std::string::npos && gt != std::string::npos && lt < gt)
            ty = ty.substr(0, lt) + "::<Frame<" + ty.substr(lt + 1, gt - lt - 1) + ">>" + ty.substr(gt + 1);
        o << "    " << n.name << " = " << (n.array_len ? "[" : "") << ty << "(";
        for (size_t i = 0; i < n.args.size(); ++i)
            o << (i ? ", " : "") << (i < n.raw_args.size() && !n.raw_args[i].empty() ? n.raw_args[i] : num(n.args[i]));
        o << ")";
        if (n.array_len) o << "; " << n.array_len << "]";
        if (n.rate_factor > 1) o << " * " << n.rate_factor;
        o << ";\n";
    }
    o << "}\n\nconnections {\n";
    std::set<std::string> bus;
    for (const auto& n : g.nodes)
        if (n.bus) bus.insert(n.name);
    for (size_t ei = 0; ei < g.edges.size(); ++ei) {
        const GEdge& e = g.edges[ei];
        auto root = [](const std::string& s) { return s.substr(0, s.find('.')); };
        if (ei + 1 < g.edges.size() && g.edges[ei + 1].feedback && !e.feedback) { // `src -> [via] -> dst`
            const GEdge& f = g.edges[ei + 1];
            const std::string via = root(f.src);
            if (e.dst == via + ".input" && f.src == via + ".output") {
                std::string label = via;
                for (const auto& n : g.nodes)
                    if (n.name == via && via.rfind("__inline_delay_", 0) == 0 && !n.args.empty())
                        label = std::to_string((unsigned)n.args[0]);
                o << "    " << (e.policy.empty() ? "" : "[" + e.policy + "] ") << e.src << " -> [" << label << "] -> " << f.dst
                  << ";\n";
                ++ei;
                continue;
            }
        }
        if (e.feedback) throw std::runtime_error("oscen graph dsl: unpaired feedback edge '" + e.src + " -> " + e.dst + "'");
        if (bus.count(root(e.src)) || bus.count(root(e.dst))) {
            o << "    // post-mix: " << e.src << " -> " << e.dst << ";\n";
            continue;
        }
        o << "    " << (e.policy.empty() ? "" : "[" + e.policy + "] ") << e.src << " -> " << e.dst << ";\n";
    }
    o << "}\n";
    return o.str();
}

} // namespace ogc
