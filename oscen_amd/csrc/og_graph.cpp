// og_graph.cpp -- graph lowering + HIP code generation (see og_graph.h).
#include "og_graph.h"
#include "og_rt_digest.h" // OG_RT_DIGEST: digest of the device headers the generated kernels include

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <set>
#include <sstream>
#include <stdexcept>

namespace ogc {
std::string normalize_type(const std::string& type);

uint64_t fnv1a(const std::string& s)
{
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

int CompiledGraph::find_input(const std::string& n) const
{
    for (size_t i = 0; i < inputs.size(); ++i)
        if (inputs[i].decl.name == n) return (int)i;
    return -1;
}

namespace {

[[noreturn]] void fail(const std::string& m) { throw std::runtime_error("oscen graph: " + m); }
// a well-formed graph that uses something this build lacks: carries OG_E_UNSUPPORTED (og_abi.h)
[[noreturn]] void fail_unsupported(const std::string& m) { throw ogabi::Unsupported("oscen graph: " + m); }

std::string flit(float v)
{ // exact float literal
    if (std::isinf(v)) return v > 0 ? "__builtin_inff()" : "(-__builtin_inff())";
    char buf[64];
    snprintf(buf, sizeof buf, "%af", (double)v);
    return buf;
}
uint32_t fbits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

Rate join(Rate a, Rate b)
{
    if (a == Rate::Vary || b == Rate::Vary) return Rate::Vary;
    const bool voice = (a == Rate::VBlock || b == Rate::VBlock);
    const bool frame = (a == Rate::UFrame || b == Rate::UFrame);
    if (voice && frame) return Rate::Vary;
    if (voice) return Rate::VBlock;
    if (frame) return Rate::UFrame;
    if (a == Rate::UBlock || b == Rate::UBlock) return Rate::UBlock;
    return Rate::Const;
}

struct Val {
    std::string e;
    Rate rate = Rate::Const;
    HostFn host;                  // available when rate <= UBlock
    std::set<int> voice_inputs;   // per-voice value inputs this depends on (rate VBlock)
    bool inner = false;           // defined inside the oversampled (x N) inner loop
    bool lane = false;            // one value per lane of an LPV > 1 voice (an `[f32; 32]` endpoint)
    bool stream = false;          // a graph-level stream input: voice-uniform, but a signal (resampled across rate domains)
    std::vector<Val> ch;          // a Frame<N> payload (oscen-lib/src/frame.rs): its N channels, each a scalar value; `e` unused
    bool is_frame() const { return !ch.empty(); }
};

Val vconst(float c)
{
    Val v;
    v.e = flit(c);
    v.rate = Rate::Const;
    v.host = [c](const UEnv&) { return c; };
    return v;
}

// ---- expression parsing ------------------------------------------------------
struct Expr {
    // Call: `half(a.output)`, `dsp::decode_ms(s.output)`, `Frame::<2>(a.output, b.output)` (ast.rs:126-128);
    // Method: `x.tanh()`, `x.clamp(0.0, 1.0)` (ast.rs:120-121); Chan: `s.output[1]` (one channel of a Frame<N>)
    enum T { Num, Ref, Bin, Neg, Call, Method, Chan } t = Num;
    float num = 0;
    std::string node, port; // Ref: port empty => bare identifier.  Call: node = the path as written; Method: node = its name
    char op = 0;
    std::shared_ptr<Expr> a, b;
    std::vector<std::shared_ptr<Expr>> args; // Call / Method
    long index = 0;                          // Chan
};
using ExprP = std::shared_ptr<Expr>;

struct Parser {
    const std::string& s;
    size_t i = 0;
    explicit Parser(const std::string& str) : s(str) {}
    void ws()
    {
        while (i < s.size() && isspace((unsigned char)s[i])) ++i;
    }
    bool eat(char c)
    {
        ws();
        if (i < s.size() && s[i] == c) {
            ++i;
            return true;
        }
        return false;
    }
    std::string ident()
    {
        ws();
        size_t b = i;
        while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_')) ++i;
        return s.substr(b, i - b);
    }
    ExprP primary()
    {
        ws();
        if (i >= s.size()) fail("unexpected end of expression '" + s + "'");
        if (eat('(')) {
            ExprP e = sum();
            if (!eat(')')) fail("missing ')' in '" + s + "'");
            return postfix(e);
        }
        if (eat('-')) {
            auto e = std::make_shared<Expr>();
            e->t = Expr::Neg;
            e->a = primary();
            return e;
        }
        if (isdigit((unsigned char)s[i]) || s[i] == '.') {
            std::string num;
            while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '.' || s[i] == '_' ||
                                    ((s[i] == '+' || s[i] == '-') && i > 0 && (s[i - 1] == 'e' || s[i - 1] == 'E')))) {
                if (s[i] != '_') num.push_back(s[i]);
                ++i;
            }
            // strip rust suffixes
            for (const char* suf : {"f32", "f64"}) {
                size_t p = num.rfind(suf);
                if (p != std::string::npos && p + 3 == num.size()) num.erase(p);
            }
            auto e = std::make_shared<Expr>();
            e->t = Expr::Num;
            e->num = strtof(num.c_str(), nullptr);
            return e;
        }
        std::string id = ident();
        if (id.empty()) fail("bad expression '" + s + "'");
        // a path `segment (:: segment)*` names a function (parse.rs:1046-1084); `::<N>` is the dropped width of a Frame
        // constructor and is rejected on any other path
        std::string path = id, last = id;
        size_t n_seg = 1;
        for (;;) {
            ws();
            if (!(i + 1 < s.size() && s[i] == ':' && s[i + 1] == ':')) break;
            i += 2;
            ws();
            if (i < s.size() && s[i] == '<') {
                if (last != "Frame") fail("turbofish arguments are only supported on the `Frame` constructor ('" + s + "')");
                while (i < s.size() && s[i] != '>') ++i;
                if (i >= s.size()) fail("missing '>' in '" + s + "'");
                ++i;
                ws();
                if (!(i < s.size() && s[i] == '(')) fail("expected `(` after `Frame::<N>` in '" + s + "'");
                break;
            }
            last = ident();
            if (last.empty()) fail("bad path in '" + s + "'");
            path += "::" + last;
            n_seg += 1;
        }
        ws();
        if (i < s.size() && s[i] == '(') {
            // `name(` directly after a bare identifier or a path: a call -- but not the legacy `osc.output()` accessor,
            // which is handled after the port below
            ++i;
            auto c = std::make_shared<Expr>();
            c->t = Expr::Call;
            c->node = path;
            c->port = last;
            c->args = call_args();
            return postfix(c);
        }
        if (n_seg > 1)
            fail("a path is only valid as a function name in a connection; endpoints use `node.field` ('" + s + "')");
        auto e = std::make_shared<Expr>();
        e->t = Expr::Ref;
        e->node = id;
        ws();
        if (i + 1 < s.size() && s[i] == '.' && (isalpha((unsigned char)s[i + 1]) || s[i + 1] == '_')) {
            const size_t save = i;
            ++i;
            const std::string port = ident();
            ws();
            if (i < s.size() && s[i] == '(' && !(i + 1 < s.size() && s[i + 1] == ')' && !is_method(port))) {
                i = save; // `input.tanh()`: a method on a bare identifier (graph input)
            } else {
                e->port = port;
                if (eat('(')) { // legacy `osc.output()` accessor syntax
                    if (!eat(')')) fail("bad accessor in '" + s + "'");
                }
            }
        }
        return postfix(e);
    }
    static bool is_method(const std::string& name);
    std::vector<ExprP> call_args() // after '(' : `expr (, expr)*` up to the matching ')'
    {
        std::vector<ExprP> args;
        ws();
        if (eat(')')) return args;
        for (;;) {
            args.push_back(sum());
            if (eat(',')) {
                ws();
                if (eat(')')) break; // trailing comma
                continue;
            }
            if (!eat(')')) fail("missing ')' in '" + s + "'");
            break;
        }
        return args;
    }
    // `[k]` (a channel of a Frame<N>) and `.method(args)` after a primary (parse.rs:1098-1123)
    ExprP postfix(ExprP e)
    {
        for (;;) {
            ws();
            if (i < s.size() && s[i] == '[') {
                ++i;
                ws();
                std::string num;
                while (i < s.size() && isdigit((unsigned char)s[i])) num.push_back(s[i++]);
                if (num.empty() || !eat(']')) fail("expected `[index]` in '" + s + "'");
                auto c = std::make_shared<Expr>();
                c->t = Expr::Chan;
                c->a = e;
                c->index = strtol(num.c_str(), nullptr, 10);
                e = c;
            } else if (i + 1 < s.size() && s[i] == '.' && (isalpha((unsigned char)s[i + 1]) || s[i + 1] == '_')) {
                ++i;
                const std::string name = ident();
                if (!eat('(')) fail("'" + name + "' in '" + s + "': a field can only follow a node name; a method call needs `(`");
                auto m = std::make_shared<Expr>();
                m->t = Expr::Method;
                m->node = name;
                m->a = e;
                m->args = call_args();
                e = m;
            } else {
                return e;
            }
        }
    }
    ExprP product()
    {
        ExprP l = primary();
        for (;;) {
            ws();
            if (i < s.size() && (s[i] == '*' || s[i] == '/')) {
                char op = s[i++];
                auto e = std::make_shared<Expr>();
                e->t = Expr::Bin;
                e->op = op;
                e->a = l;
                e->b = primary();
                l = e;
            } else
                return l;
        }
    }
    ExprP sum()
    {
        ExprP l = product();
        for (;;) {
            ws();
            if (i < s.size() && (s[i] == '+' || s[i] == '-')) {
                char op = s[i++];
                auto e = std::make_shared<Expr>();
                e->t = Expr::Bin;
                e->op = op;
                e->a = l;
                e->b = product();
                l = e;
            } else
                return l;
        }
    }
    ExprP parse()
    {
        ExprP e = sum();
        ws();
        if (i != s.size()) fail("trailing characters in '" + s + "'");
        return e;
    }
};

void collect_refs(const ExprP& e, std::vector<const Expr*>& out)
{
    if (!e) return;
    if (e->t == Expr::Ref) out.push_back(e.get());
    collect_refs(e->a, out);
    collect_refs(e->b, out);
    for (const ExprP& x : e->args) collect_refs(x, out);
}

std::string sanitize(const std::string& s);
bool is_ident(const std::string& s);

// ---- f32 methods usable on a connection (`x.tanh()`, `x.clamp(0.0, 1.0)`: ast.rs:120-121; the reference passes
// the call through to Rust's f32, whose transcendental methods bind to the platform libm) ------------------------
struct MethodInfo {
    int n_args;
    const char* dev; // device expression: $0 = receiver, $1.. = arguments
    float (*host)(float, float, float);
};
const std::map<std::string, MethodInfo>& method_table()
{
    static const std::map<std::string, MethodInfo> T = {
        {"abs", {0, "__builtin_fabsf($0)", [](float x, float, float) { return fabsf(x); }}},
        {"sqrt", {0, "__fsqrt_rn($0)", [](float x, float, float) { return sqrtf(x); }}},
        {"cbrt", {0, "cbrtf($0)", [](float x, float, float) { return cbrtf(x); }}},
        {"recip", {0, "(1.0f / $0)", [](float x, float, float) { return 1.0f / x; }}},
        {"tanh", {0, "tanhf($0)", [](float x, float, float) { return tanhf(x); }}},
        {"sinh", {0, "sinhf($0)", [](float x, float, float) { return sinhf(x); }}},
        {"cosh", {0, "coshf($0)", [](float x, float, float) { return coshf(x); }}},
        {"sin", {0, "og_sinf_exact($0)", [](float x, float, float) { return sinf(x); }}},
        {"cos", {0, "og_cosf_exact($0)", [](float x, float, float) { return cosf(x); }}},
        {"tan", {0, "tanf($0)", [](float x, float, float) { return tanf(x); }}},
        {"asin", {0, "asinf($0)", [](float x, float, float) { return asinf(x); }}},
        {"acos", {0, "acosf($0)", [](float x, float, float) { return acosf(x); }}},
        {"atan", {0, "atanf($0)", [](float x, float, float) { return atanf(x); }}},
        {"atan2", {1, "atan2f($0, $1)", [](float x, float y, float) { return atan2f(x, y); }}},
        {"hypot", {1, "hypotf($0, $1)", [](float x, float y, float) { return hypotf(x, y); }}},
        {"exp", {0, "expf($0)", [](float x, float, float) { return expf(x); }}},
        {"exp2", {0, "exp2f($0)", [](float x, float, float) { return exp2f(x); }}},
        {"exp_m1", {0, "expm1f($0)", [](float x, float, float) { return expm1f(x); }}},
        {"ln", {0, "logf($0)", [](float x, float, float) { return logf(x); }}},
        {"ln_1p", {0, "log1pf($0)", [](float x, float, float) { return log1pf(x); }}},
        {"log2", {0, "log2f($0)", [](float x, float, float) { return log2f(x); }}},
        {"log10", {0, "log10f($0)", [](float x, float, float) { return log10f(x); }}},
        {"powf", {1, "powf($0, $1)", [](float x, float y, float) { return powf(x, y); }}},
        {"powi", {1, "powf($0, $1)", [](float x, float y, float) { return powf(x, y); }}},
        {"floor", {0, "floorf($0)", [](float x, float, float) { return floorf(x); }}},
        {"ceil", {0, "ceilf($0)", [](float x, float, float) { return ceilf(x); }}},
        {"round", {0, "roundf($0)", [](float x, float, float) { return roundf(x); }}},
        {"trunc", {0, "truncf($0)", [](float x, float, float) { return truncf(x); }}},
        {"fract", {0, "($0 - truncf($0))", [](float x, float, float) { return x - truncf(x); }}},
        {"signum", {0, "(($0) != ($0) ? ($0) : __builtin_copysignf(1.0f, ($0)))", [](float x, float, float) { return x != x ? x : (std::signbit(x) ? -1.0f : 1.0f); }}},
        {"min", {1, "fminf($0, $1)", [](float x, float y, float) { return fminf(x, y); }}},
        {"max", {1, "fmaxf($0, $1)", [](float x, float y, float) { return fmaxf(x, y); }}},
        {"clamp", {2, "(($0) != ($0) ? ($0) : fminf(fmaxf($0, $1), $2))", [](float x, float lo, float hi) { return x != x ? x : (x < lo ? lo : (x > hi ? hi : x)); }}},
        {"mul_add", {2, "__builtin_fmaf($0, $1, $2)", [](float x, float a, float b) { return fmaf(x, a, b); }}},
        {"to_radians", {0, "($0 * 0x1.1df46ap-6f)", [](float x, float, float) { return x * 0x1.1df46ap-6f; }}},
        {"to_degrees", {0, "($0 * 0x1.ca5dc2p+5f)", [](float x, float, float) { return x * 0x1.ca5dc2p+5f; }}},
    };
    return T;
}
bool Parser::is_method(const std::string& name) { return method_table().count(name) > 0; }

// ---- named functions applied on a connection (og_register_function): the pure in-scope Rust functions of
// `decode_ms(s.output) -> out`, `dsp::half(a.output) -> out` (oscen-lib/tests/connection_expr_functions.rs) ----------
std::map<std::string, UserFunction>& function_registry()
{
    static std::map<std::string, UserFunction> R;
    return R;
}
const UserFunction* lookup_function(const std::string& path, const std::string& last)
{
    auto& R = function_registry();
    auto it = R.find(path); // as written (`dsp::decode_ms`), then by its last segment (what `use dsp::decode_ms` would give)
    if (it == R.end()) it = R.find(last);
    if (it == R.end())
        for (auto& kv : R) { // a registration under a longer path matches a call through a shorter one
            const std::string& k = kv.first;
            if (k.size() > last.size() + 2 && k.compare(k.size() - last.size() - 2, std::string::npos, "::" + last) == 0) return &kv.second;
        }
    return it == R.end() ? nullptr : &it->second;
}

// ---- node type registry ------------------------------------------------------
struct PortSpec {
    const char* name;
    Kind kind;
    float def;
    int arg; // ctor argument that overrides the default, or -1
    int channels = 1; // stream ports: 1 = f32, N = Frame<N>
};

struct Codegen;
struct NodeCtx;
using Emitter = void (*)(NodeCtx&);

struct NodeTypeInfo {
    std::vector<PortSpec> inputs;
    std::vector<const char*> outputs;
    Emitter emit;
    int variant;
    size_t nargs;
    int lpv = 1; // lanes per voice this node type needs (32: per-harmonic arrays)
    const UserNodeType* user = nullptr; // registered through og_register_node (emit_user)
    std::vector<std::string> ev_outputs; // `#[output(event)]` fields (user node types only)
    std::vector<int> out_channels;       // per output: 1 = f32, N = Frame<N> (empty: all f32)
};

const NodeTypeInfo* lookup_type(const std::string& type);
int user_weight(const std::string& type);

// rough VALU cost per tick of a node type (used to balance the two-stage split)
int node_weight(const std::string& type)
{
    // (round 5, static counts of the quiet chunk bodies: profiles/r05g_session6.log picked the cut these weights give)
    if (type.rfind("AdsrEnvelope", 0) == 0) return 5;  // tolerance mode: sub, fma, countdown share (+ release: sub, cvt, rcp, fma)
    if (type.rfind("FmOperator", 0) == 0) return 10;   // fma, add, v_sin_f32 (a quarter-rate instruction: 4), 2 mul, add, v_fract
    if (type.rfind("TptFilter", 0) == 0) return 25;
    if (type.rfind("PolyBlepOscillator", 0) == 0) return 30;
    if (type.rfind("Oscillator", 0) == 0) return 22;
    if (type.rfind("IirLowpass", 0) == 0) return 14;
    if (type.rfind("LP18Filter", 0) == 0) return 40;
    if (type.rfind("Delay", 0) == 0) return 30;
    if (type.rfind("Crossfade", 0) == 0) return 2;
    if (type.rfind("HardClip", 0) == 0) return 2;
    const int uw = user_weight(type);
    return uw > 0 ? uw : 1;
}

struct NodeInst {
    const GNode* decl = nullptr;
    const NodeTypeInfo* type = nullptr;
    int id = -1;
    bool live = true;
    struct Src {
        ExprP e;
        std::string policy;
    };
    std::map<std::string, std::vector<Src>> in_edges; // stream/value port -> sources in edge order
    int domain = 0; // 0 = outer before the inner loop, 1 = oversampled inner loop, 2 = outer after it
    // event port -> its source.  An event edge is clear + copy (graph/static_context.rs:84-155), so on fan-in only the
    // LAST connected source delivers: at most one of the two maps holds the port, ev_edges with a single entry.
    std::map<std::string, std::vector<int>> ev_edges;                     // ... a graph event input (index)
    std::map<std::string, std::pair<int, std::string>> ev_node_edges;     // ... an event output of another node (node, port)
};

struct Codegen {
    const GraphDesc& g;
    CompiledGraph& out;
    std::vector<NodeInst> nodes;
    std::map<std::string, int> node_by_name, input_by_name, output_by_name;
    std::map<std::string, Val> output_vals; // graph outputs that other outputs read: their value on the current frame
    std::map<std::string, Val> node_outputs; // "n<id>.<port>" -> value
    std::set<std::string> fb_sources;            // "n<id>.<port>" fed into a feedback edge
    std::map<std::string, std::string> fb_vars;  // ... -> state variable holding last frame's value
    std::vector<char> emitted;                   // per node

    // Emitted code sections.  There are up to four sets: a graph may be cut into pipeline stages
    // (see "pipeline stages" in compile()); nodes of stage s write into sec[s].  The ordinary
    // kernel simply concatenates the sets; the pipelined kernels give groups of them to separate waves.
    struct Sect {
        std::ostringstream decl, load, derive, pre, pre_store, store;
        std::ostringstream chunk_begin; // top of every OG_BUS_CHUNK-frame chunk (delay-line staging)
        std::vector<std::string> env_cnts; // countdowns of this stage's envelopes (a chunk in which none of them
                                           // reaches 0 runs the tick without the stage-end checks)
        std::vector<std::string> env_rs;   // their release flags (1.0f in Release): no lane releasing -> no release arithmetic
        std::vector<std::string> fast_conds; // per-lane conditions under which this stage's nodes may run their
                                             // `steady` tick for a whole chunk of CHUNK frames (e.g. og::ep_amp_tick)
        // per-frame code, multirate layout of emit_frame.rs:114-176:
        //   s_pre (outer nodes) | s_up (upsamplers) | for j<N { s_inner ; s_cap } | s_down (downsamplers) | s_post
        std::ostringstream s_pre, s_up, s_inner, s_log, s_cap, s_down, s_post; // (s_log: graph event outputs of inner nodes, logged before s_cap clears them)
        // step 6a of the reference's multirate body (codegen/emit_frame.rs:150-160): the event handlers of an INNER node fed
        // by another inner node run once per OUTER tick, in front of the inner loop, over everything the producer pushed
        // during the previous outer tick; the producer's queue is cleared there, not per inner tick
        std::ostringstream s_ev6a, s_ev6a_clear;
        std::map<int, std::ostringstream> ev_handlers; // per graph event input
    };
    Sect sec[16];
    int cs = 0;  // stage being emitted
    int dom = 0; // rate domain being emitted: 0 pre, 1 inner, 2 post
    Sect& S() { return sec[cs]; }
    std::ostringstream& os() { return dom == 0 ? S().s_pre : (dom == 1 ? S().s_inner : S().s_post); }
    std::ostringstream common_decl, common_load; // per-voice value inputs: visible to both stages
    std::map<std::string, std::string> user_fns; // device functions of the user node types this graph uses
    int blk_slot0 = -1;  // first of the 32 slots holding the block starts of a launch (allocated when a handler reads frame_offset)
    int ev_capacity = 2; // OG_NODE_EVENTS_PER_FRAME of this graph: the largest event_queue_capacity among its node types
    std::set<std::pair<int, std::string>> inner_ev_sources; // (node id, event output) read by another node of the oversampled region
    std::set<std::pair<int, std::string>> outer_ev_sources; // (inner node id, event output) read by an outer node behind the region
    int N = 1;        // oversampling factor of the `* N` nodes (1 = none)
    int n_cross = 0;  // cross-rate edges emitted so far
    bool any_derive = false;
    // pipeline bookkeeping
    bool split = false;
    int n_stages = 1;
    std::vector<std::vector<int>> groups2, groups4; // stages of every wave of the 2- / 4-wave pipelines (or empty)
    std::vector<int> stage_of; // per node
    struct XVal {              // a node output read by a later stage
        std::string var, alias;
        int from;
        std::set<int> users; // consuming stages
    };
    std::vector<XVal> xvals;
    bool dynamic_events = false; // some node receives events from another node: they arrive unannounced, so the
                                 // chunk variants that rely on "nothing happens in this chunk" and the pipelines are off
    std::ostringstream frame_end; // end of every frame: clear_event_outputs()
    std::ostringstream frame_log; // ... right before it: events of the graph's event outputs go to the host log
    bool bus_all_lanes = false; // every lane of a multi-lane voice contributes a share to the mix bus (og::ep_bank_tick)
    // envelopes whose stage-end fix-up has not been emitted yet: (countdown expression, fix-up code).
    // Flushed as ONE wave-uniform check before the next node that is not an envelope (which may read
    // their outputs), before the stage changes and before the graph output is formed.
    std::vector<std::pair<std::string, std::string>> pending_post;
    void flush_post()
    {
        if (pending_post.empty()) return;
        std::string m = pending_post[0].first;
        for (size_t i = 1; i < pending_post.size(); ++i) m = "min(" + m + ", " + pending_post[i].first + ")";
        os() << "        if constexpr (decltype(chk)::value) { // (compiled out of chunks in which no countdown can reach 0)\n"
             << "        if (__any((int)(" << m << " == 0u))) { // rare per-voice work (stage ends)\n";
        for (auto& pp : pending_post) os() << pp.second;
        os() << "        }\n        }\n";
        for (auto& pp : pending_post) {
            S().env_cnts.push_back(pp.first);
            S().env_rs.push_back(pp.first.substr(0, pp.first.size() - 4) + ".rs"); // "<E>.cnt" -> "<E>.rs"
        }
        pending_post.clear();
    }

    Codegen(const GraphDesc& gd, CompiledGraph& cg) : g(gd), out(cg) {}

    int new_slot(std::function<uint32_t(const UEnv&)> fn)
    {
        int s = out.n_slots++;
        out.uprogs.push_back({s, std::move(fn)});
        return s;
    }
    int new_state(const std::string& name, bool is_float, std::function<uint32_t(const UEnv&)> init)
    {
        out.state.push_back({name, is_float, std::move(init)});
        return (int)out.state.size() - 1;
    }

    Val input_val(int idx)
    {
        const InputInfo& in = out.inputs[idx];
        Val v;
        if (in.decl.kind == Kind::Stream) { // `<stream_in>_block[f]`: the same sample for every voice of the bank
            auto row = [&](int r) {
                Val c;
                c.e = "ST(" + std::to_string(r) + ")";
                c.rate = Rate::UFrame;
                c.stream = true;
                return c;
            };
            if (in.decl.channels <= 1) return row(in.stream_row);
            v.rate = Rate::Vary; // (a frame: its channels carry the rates)
            for (int k = 0; k < in.decl.channels; ++k) v.ch.push_back(row(in.stream_row + k));
            return v;
        }
        if (in.decl.kind != Kind::Value) fail("input '" + in.decl.name + "' is not a value input");
        if (in.decl.per_voice) {
            v.e = "vin_" + std::to_string(idx);
            v.rate = Rate::VBlock;
            v.voice_inputs.insert(idx);
        } else if (in.ramp_row >= 0) {
            v.e = "RV(" + std::to_string(in.ramp_row) + ", " + std::to_string(in.slot) + ")";
            v.rate = Rate::UFrame;
        } else {
            v.e = "SF(" + std::to_string(in.slot) + ")";
            v.rate = Rate::UBlock;
            v.host = [idx](const UEnv& e) { return e.input_values[idx]; };
        }
        return v;
    }

    // Bring a source value into the destination's rate domain (ir/lower.rs:824-906 kernel
    // selection; dispatch/stream.rs:94-104): same rate -> as is; value-kind or voice-uniform
    // source into an oversampled node -> Latch (the value is simply visible in the inner loop);
    // stream outer -> inner = Up{N, policy}; inner -> outer = Down{N, policy}; default policy Sinc.
    Val cross(Val v, const std::string& policy, bool dst_inner, bool value_port)
    {
        if (v.is_frame()) {
            // The reference's resamplers are generic over F: AudioFrame and work channel by channel with one state per
            // channel (resample/sinc_fir.rs:96-144; `Frame<2>` == scalar per channel, tests/resample_kernels.rs:475-626):
            // a frame edge crosses a rate boundary as its channels do.  The edge's latency counts once.
            Val r;
            r.rate = Rate::Vary;
            const uint32_t lat0 = out.latency_samples;
            uint32_t lat1 = lat0;
            for (const Val& c : v.ch) {
                out.latency_samples = lat0;
                r.ch.push_back(cross(c, policy, dst_inner, value_port));
                lat1 = std::max(lat1, out.latency_samples);
            }
            out.latency_samples = lat1;
            return r;
        }
        if (N <= 1 || v.inner == dst_inner) {
            // (a policy on a same-rate edge is accepted and ignored, like the reference: classify_edge_ir
            //  `(Same, Same) => EdgeKernel::None` whatever the policy, ir/lower.rs:878 -- the `_1x` variant of
            //  oversample_variants! keeps its `[sinc]` edge, oscen-lib/tests/oversample_variants.rs:8-23)
            return v;
        }
        const std::string pol = policy.empty() ? "sinc" : policy;
        if (pol != "sinc" && pol != "sinc_iir" && pol != "linear" && pol != "latch")
            fail("unknown connection policy [" + policy + "]");
        const std::string id = std::to_string(n_cross++);
        const std::string NN = std::to_string(N);
        const int stages = N >= 8 ? 3 : (N >= 4 ? 2 : 1);
        auto state_arr = [&](const std::string& var, int rows, int cols) {
            S().decl << "    float " << var << "[" << rows << "][" << cols << "] = {};\n";
            for (int r = 0; r < rows; ++r)
                for (int c2 = 0; c2 < cols; ++c2) {
                    int w = new_state("edge" + id + "." + var + "[" + std::to_string(r) + "][" + std::to_string(c2) + "]",
                                      true, [](const UEnv&) { return 0u; });
                    S().load << "        " << var << "[" << r << "][" << c2 << "] = og::ld_f(A, c, " << w << ");\n";
                    S().store << "        og::st_f(A, c, " << w << ", " << var << "[" << r << "][" << c2 << "]);\n";
                }
        };
        Val r;
        r.rate = Rate::Vary;
        if (dst_inner) { // Up edge (emit_frame.rs:254-307): upsample once per outer frame into a [N] buffer
            if ((v.rate != Rate::Vary && !v.stream) || value_port || pol == "latch") { // Latch: every inner tick sees the outer value
                v.inner = true;
                return v;
            }
            const std::string buf = "up" + id, st = "up" + id + "_st";
            S().s_up << "        float " << buf << "[" << NN << "];\n";
            if (pol == "sinc") {
                state_arr(st, stages, 12);
                S().s_up << "        og::sinc_up<" << NN << ">(" << st << ", " << v.e << ", " << buf << ");\n";
            } else if (pol == "sinc_iir") {
                state_arr(st, stages, 9);
                S().s_up << "        og::iir_up<" << NN << ">(" << st << ", " << v.e << ", " << buf << ");\n";
            } else {
                state_arr(st, 1, 1);
                S().s_up << "        og::linear_up<" << NN << ">(" << st << "[0][0], " << v.e << ", " << buf << ");\n";
            }
            r.e = buf + "[j]";
            r.inner = true;
            return r;
        }
        // Down edge (emit_frame.rs:474-514): capture every inner tick, downsample once per outer frame
        const std::string buf = "dn" + id, st = "dn" + id + "_st";
        S().s_up << "        float " << buf << "[" << NN << "];\n";
        S().s_cap << "            " << buf << "[j] = " << v.e << ";\n";
        int lat = 0;
        if (pol == "sinc") {
            state_arr(st, stages, 24);
            S().s_down << "        const float " << buf << "_o = og::sinc_down<" << NN << ">(" << st << ", " << buf << ");\n";
            lat = 11 * (N - 1);
        } else if (pol == "sinc_iir") {
            state_arr(st, stages, 9);
            S().s_down << "        const float " << buf << "_o = og::iir_down<" << NN << ">(" << st << ", " << buf << ");\n";
            lat = 2 * (N - 1);
        } else if (pol == "linear") {
            S().s_down << "        const float " << buf << "_o = og::linear_down<" << NN << ">(" << buf << ");\n";
            lat = (N - 1) / 2;
        } else {
            S().s_down << "        const float " << buf << "_o = " << buf << "[0];\n";
        }
        out.latency_samples += (uint32_t)(lat / N); // emit_struct.rs:534-570
        r.e = buf + "_o";
        r.inner = false;
        return r;
    }

    // scalar arithmetic of a compound source (`a.x * b.y`); op 'n' = unary minus of a
    Val arith(const Val& a, const Val& b, char op)
    {
        Val r;
        if (op == 'n') {
            r.e = "(-" + a.e + ")";
            r.rate = a.rate;
            r.inner = a.inner;
            r.stream = a.stream;
            r.voice_inputs = a.voice_inputs;
            if (a.host) {
                HostFn h = a.host;
                r.host = [h](const UEnv& env) { return -h(env); };
            }
            return r;
        }
        r.e = "(" + a.e + " " + op + " " + b.e + ")";
        r.rate = join(a.rate, b.rate);
        r.stream = a.stream || b.stream;
        if (a.inner != b.inner && a.rate == Rate::Vary && b.rate == Rate::Vary)
            fail("expression mixes outer-rate and oversampled node outputs; connect them through a cross-rate edge");
        r.inner = a.inner || b.inner;
        r.voice_inputs = a.voice_inputs;
        r.voice_inputs.insert(b.voice_inputs.begin(), b.voice_inputs.end());
        if (a.host && b.host && r.rate <= Rate::UBlock) {
            HostFn ha = a.host, hb = b.host;
            r.host = [ha, hb, op](const UEnv& env) {
                float x = ha(env), y = hb(env);
                switch (op) {
                case '+': return x + y;
                case '-': return x - y;
                case '*': return x * y;
                default: return x / y;
                }
            };
        }
        return r;
    }
    // Frame<N> arithmetic (oscen-lib/src/frame.rs:104-134): Add / Sub element-wise between frames of one width,
    // Mul<f32> = frame * scalar, Neg; nothing else exists on the reference's AudioFrame either
    Val frame_arith(const Val& a, const Val& b, char op, const std::string& what)
    {
        Val r;
        r.rate = Rate::Vary;
        if (a.is_frame() && b.is_frame()) {
            if (a.ch.size() != b.ch.size()) fail("frames of different widths in '" + what + "'");
            if (op != '+' && op != '-') fail("Frame<N> supports frame + frame, frame - frame and frame * f32 ('" + what + "')");
            for (size_t i = 0; i < a.ch.size(); ++i) r.ch.push_back(arith(a.ch[i], b.ch[i], op));
        } else if (a.is_frame() && op == '*') {
            for (const Val& c : a.ch) r.ch.push_back(arith(c, b, '*'));
        } else {
            fail("Frame<N> supports frame + frame, frame - frame and frame * f32 ('" + what + "')");
        }
        return r;
    }
    std::map<std::string, int> node_frame_outputs; // "n<id>.<port>" -> N for Frame<N> outputs (channels live as "<port>#i")

    // one scalar output of a node (a whole f32 output, or channel "<port>#i" of a frame output)
    Val node_output(int ni, const std::string& node_name, const std::string& port)
    {
        const std::string key = "n" + std::to_string(ni) + "." + port;
        auto vit = node_outputs.find(key);
        if (vit == node_outputs.end() && fb_sources.count(key) && !emitted[ni] && nodes[ni].live) {
            // feedback edge whose producer runs later in the frame: the consumer sees the field the
            // producer wrote on the previous frame (the struct field persists, codegen/emit_node.rs)
            if ((nodes[ni].domain == 1) != (dom == 1))
                fail_unsupported("feedback edge from '" + node_name + "' crosses the oversampled region: both ends of a feedback edge must "
                     "run at the same rate in this version");
            auto fit = fb_vars.find(key);
            if (fit == fb_vars.end()) {
                const std::string var = "n" + std::to_string(ni) + "_" + port + "_z";
                int w = new_state(node_name + "." + port, true, [](const UEnv&) { return 0u; });
                S().decl << "    float " << var << " = 0.0f;\n";
                S().load << "        " << var << " = og::ld_f(A, c, " << w << ");\n";
                S().store << "        og::st_f(A, c, " << w << ", " << var << ");\n";
                fit = fb_vars.emplace(key, var).first;
            }
            Val v;
            v.e = fit->second;
            v.rate = Rate::Vary;
            v.inner = nodes[ni].domain == 1; // last tick's value, at the producer's rate
            return v;
        }
        if (vit == node_outputs.end())
            fail("node '" + node_name + "' has no output '" + port + "' (or it is read before it runs)");
        Val v = vit->second;
        if (split && cs > stage_of[ni]) { // value crosses a pipeline cut
            XVal* x = nullptr;
            for (auto& xv : xvals)
                if (xv.var == v.e) x = &xv;
            if (!x) {
                xvals.push_back({v.e, "x" + std::to_string(xvals.size()) + "_" + v.e, stage_of[ni], {}});
                x = &xvals.back();
            }
            x->users.insert(cs);
            v.e = x->alias;
        }
        return v;
    }

    Val eval(const ExprP& e)
    {
        switch (e->t) {
        case Expr::Num: return vconst(e->num);
        case Expr::Neg: {
            Val a = eval(e->a);
            if (!a.is_frame()) return arith(a, a, 'n');
            Val r;
            r.rate = Rate::Vary;
            for (const Val& c : a.ch) r.ch.push_back(arith(c, c, 'n'));
            return r;
        }
        case Expr::Ref: {
            if (e->port.empty()) {
                auto it = input_by_name.find(e->node);
                if (it != input_by_name.end()) return input_val(it->second);
                // a graph OUTPUT read as a source (`out_a + out_b -> out`, tests/multirate_graph.rs:444-458): the value
                // this frame's edges put into it
                auto ov = output_vals.find(e->node);
                if (ov != output_vals.end()) return ov->second;
                if (output_by_name.count(e->node))
                    fail("graph output '" + e->node + "' is read before it is formed (only another graph output may read it)");
                fail("unknown source '" + e->node + "'");
            }
            auto nit = node_by_name.find(e->node);
            if (nit == node_by_name.end()) fail("unknown node '" + e->node + "'");
            int width = 1;
            const NodeTypeInfo* ti = nodes[nit->second].type;
            for (size_t o = 0; ti && o < ti->outputs.size() && o < ti->out_channels.size(); ++o)
                if (e->port == ti->outputs[o]) width = ti->out_channels[o];
            if (width <= 1) return node_output(nit->second, e->node, e->port);
            if (fb_sources.count("n" + std::to_string(nit->second) + "." + e->port))
                fail("a feedback edge carries an f32 stream ('" + e->node + "." + e->port + "' is a Frame<" + std::to_string(width) + ">)");
            Val r;
            r.rate = Rate::Vary;
            for (int c = 0; c < width; ++c) r.ch.push_back(node_output(nit->second, e->node, e->port + "#" + std::to_string(c)));
            return r;
        }
        case Expr::Chan: { // `s.output[1]`: one channel of a Frame<N> source
            Val a = eval(e->a);
            if (!a.is_frame()) fail("`[" + std::to_string(e->index) + "]` needs a Frame<N> source (an element of a node array is `name[i].port`)");
            if (e->index < 0 || (size_t)e->index >= a.ch.size())
                fail("channel " + std::to_string(e->index) + " of a Frame<" + std::to_string(a.ch.size()) + ">");
            return a.ch[(size_t)e->index];
        }
        case Expr::Method: return method(e);
        case Expr::Call: return call(e);
        default: {
            Val a = eval(e->a), b = eval(e->b);
            if (a.is_frame() || b.is_frame()) return frame_arith(a, b, e->op, std::string("frame ") + e->op + " ...");
            return arith(a, b, e->op);
        }
        }
    }

    // joins what a derived value inherits from the values it is computed from
    static void inherit(Val& r, const Val& a)
    {
        r.rate = join(r.rate, a.rate);
        r.stream = r.stream || a.stream;
        r.inner = r.inner || a.inner;
        r.voice_inputs.insert(a.voice_inputs.begin(), a.voice_inputs.end());
    }
    void check_same_domain(const std::vector<Val>& vs, const std::string& what)
    {
        bool in = false, outr = false;
        for (const Val& v : vs) {
            if (v.rate != Rate::Vary) continue;
            (v.inner ? in : outr) = true;
        }
        if (in && outr) fail("'" + what + "' mixes outer-rate and oversampled node outputs; connect them through a cross-rate edge");
    }
    Val method(const ExprP& e)
    {
        auto mit = method_table().find(e->node);
        if (mit == method_table().end()) fail("unknown f32 method '." + e->node + "()' on a connection");
        const MethodInfo& mi = mit->second;
        if ((int)e->args.size() != mi.n_args)
            fail("'." + e->node + "()' takes " + std::to_string(mi.n_args) + " argument(s), " + std::to_string(e->args.size()) + " given");
        std::vector<Val> vs{eval(e->a)};
        for (const ExprP& x : e->args) vs.push_back(eval(x));
        for (const Val& v : vs)
            if (v.is_frame()) fail("'." + e->node + "()' is an f32 method: a Frame<N> has no such method (take a channel: `x[0]`)");
        check_same_domain(vs, "." + e->node + "()");
        Val r;
        std::string d = mi.dev;
        for (size_t k = vs.size(); k-- > 0;) { // ($2 before $1 before $0: none is a prefix of a later one)
            const std::string key = "$" + std::to_string(k);
            for (size_t p = d.find(key); p != std::string::npos; p = d.find(key, p + vs[k].e.size())) d.replace(p, key.size(), vs[k].e);
        }
        r.e = d;
        bool hosts = true;
        for (const Val& v : vs) {
            inherit(r, v);
            hosts = hosts && (bool)v.host;
        }
        if (r.rate <= Rate::UBlock) {
            if (hosts) {
                HostFn h0 = vs[0].host, h1 = vs.size() > 1 ? vs[1].host : HostFn(), h2 = vs.size() > 2 ? vs[2].host : HostFn();
                auto fn = mi.host;
                r.host = [fn, h0, h1, h2](const UEnv& env) { return fn(h0(env), h1 ? h1(env) : 0.0f, h2 ? h2(env) : 0.0f); };
            } else {
                r.rate = Rate::Vary;
            }
        }
        return r;
    }
    Val call(const ExprP& e)
    {
        std::vector<Val> vs;
        for (const ExprP& x : e->args) vs.push_back(eval(x));
        if (e->port == "Frame") { // any call path ending in `Frame` is the frame constructor (parse.rs:1051-1075)
            if (vs.size() < 2 || vs.size() > 4) fail("Frame(..) takes 2 to 4 channels ('" + e->node + "')");
            Val r;
            r.rate = Rate::Vary;
            for (const Val& v : vs) {
                if (v.is_frame()) fail("Frame(..) takes f32 channels ('" + e->node + "')");
                r.ch.push_back(v);
            }
            check_same_domain(vs, e->node + "(..)");
            return r;
        }
        const UserFunction* f = lookup_function(e->node, e->port);
        if (!f)
            fail("unknown function '" + e->node + "' in a connection: the reference resolves it to a Rust function in scope; here it has to be "
                 "registered (og_register_function)");
        if (vs.size() != f->arg_names.size())
            fail("function '" + e->node + "' takes " + std::to_string(f->arg_names.size()) + " argument(s), " + std::to_string(vs.size()) + " given");
        std::vector<Val> flat;
        std::string args;
        for (size_t k = 0; k < vs.size(); ++k) {
            const int w = f->arg_channels[k];
            if ((vs[k].is_frame() ? (int)vs[k].ch.size() : 1) != w)
                fail("function '" + e->node + "': argument '" + f->arg_names[k] + "' is " + (w > 1 ? "a Frame<" + std::to_string(w) + ">" : "an f32") +
                     ", the source is " + (vs[k].is_frame() ? "a Frame<" + std::to_string(vs[k].ch.size()) + ">" : "an f32"));
            if (k) args += ", ";
            if (w > 1) {
                std::string init;
                for (const Val& c : vs[k].ch) {
                    init += (init.empty() ? "" : ", ") + c.e;
                    flat.push_back(c);
                }
                args += "og::Frame<" + std::to_string(w) + ">{{" + init + "}}";
            } else {
                args += vs[k].e;
                flat.push_back(vs[k]);
            }
        }
        check_same_domain(flat, e->node + "(..)");
        const std::string fn = "og_fn_" + sanitize(f->name);
        if (!user_fns.count("~fn:" + f->name)) {
            std::ostringstream d;
            auto ty = [](int w) { return w > 1 ? "og::Frame<" + std::to_string(w) + ">" : std::string("float"); };
            d << "// function " << f->name << " (og_register_function)\n__device__ __forceinline__ " << ty(f->result_channels) << " " << fn << "(";
            for (size_t k = 0; k < f->arg_names.size(); ++k)
                d << (k ? ", " : "") << "const " << ty(f->arg_channels[k]) << " " << f->arg_names[k];
            d << ")\n{\n" << f->source << "\n}\n";
            user_fns["~fn:" + f->name] = d.str();
        }
        // no host-side evaluation of device source: the result is a per-frame value whatever its arguments are
        Val base;
        base.rate = Rate::Vary;
        for (const Val& v : flat) inherit(base, v);
        base.rate = Rate::Vary;
        const std::string c = fn + "(" + args + ")";
        if (f->result_channels <= 1) {
            base.e = c;
            return base;
        }
        Val r;
        r.rate = Rate::Vary;
        // (One textual call per channel: the value is an EXPRESSION that the caller places wherever the edge is evaluated --
        //  the frame body, the inner loop of an oversampled region, a resampler's capture slot -- so it cannot name a
        //  temporary of some other scope.  The function is force-inlined and pure by contract (og_register_function: "pure
        //  function of its arguments"), the identical calls fold into one evaluation; ADVICE r3 asked for a temporary,
        //  tried in round 4: it broke calls inside `* N` regions.)
        for (int k = 0; k < f->result_channels; ++k) {
            Val ch = base;
            ch.e = c + ".v[" + std::to_string(k) + "]";
            r.ch.push_back(ch);
        }
        return r;
    }
};

struct NodeCtx {
    Codegen& cg;
    NodeInst& n;
    std::string p; // local-name prefix "n<id>_"

    const PortSpec& port(const std::string& name) const
    {
        for (const auto& ps : n.type->inputs)
            if (name == ps.name) return ps;
        fail("node '" + n.decl->name + "' has no input '" + name + "'");
    }
    float def(const std::string& name) const
    {
        const PortSpec& ps = port(name);
        if (ps.arg >= 0 && (size_t)ps.arg < n.decl->args.size()) return n.decl->args[ps.arg];
        return ps.def;
    }
    bool connected(const std::string& name) const { return n.in_edges.count(name) > 0; }
    // Handler of event input `port`: gen(value expression) -> statements.  Events of a graph event input are applied by
    // the kernel's event loop before the frame's ticks; events another node pushed on this frame are applied right here
    // (= before this node's process(), after the producer's: process_event_inputs(), oscen-macros/src/lib.rs:265-285).
    void on_event(const std::string& port, const std::function<std::string(const std::string&)>& gen);
    // resolved value of a stream/value input: constant default, one source, or the sum of sources
    bool lane_ok = false;
    Val in_lane(const std::string& name)
    {
        lane_ok = true;
        Val v = in(name);
        lane_ok = false;
        return v;
    }
    // a Frame<N> stream input: the N channel values (unconnected: every channel holds the default)
    Val in_frame(const std::string& name, int width)
    {
        frame_ok = width;
        Val v = in(name);
        frame_ok = 1;
        if (!v.is_frame()) {
            if (n.in_edges.count(name))
                fail("node '" + n.decl->name + "': input '" + name + "' is a Frame<" + std::to_string(width) + ">, its source is an f32 stream");
            Val r;
            r.rate = Rate::Vary;
            for (int c = 0; c < width; ++c) r.ch.push_back(v);
            return r;
        }
        return v;
    }
    int frame_ok = 1;
    Val in(const std::string& name)
    {
        Val v = in_any(name);
        if (v.is_frame() && (int)v.ch.size() != frame_ok)
            fail("node '" + n.decl->name + "': input '" + name + "' " +
                 (frame_ok == 1 ? "takes an f32 stream" : "is a Frame<" + std::to_string(frame_ok) + ">") + ", its source is a Frame<" +
                 std::to_string(v.ch.size()) + ">");
        return v;
    }
    Val in_any(const std::string& name)
    {
        auto it = n.in_edges.find(name);
        if (it == n.in_edges.end()) return vconst(def(name));
        const bool dst_inner = n.domain == 1;
        const bool value_port = port(name).kind == Kind::Value;
        if (it->second.size() > 1) // fan-in sum: same-rate simple sources only (emit_node.rs:35-111)
            for (auto& src : it->second)
                if (!src.policy.empty()) fail("fan-in summing supports only same-rate sources (input '" + name + "')");
        Val acc = cg.cross(cg.eval(it->second[0].e), it->second[0].policy, dst_inner, value_port);
        if (acc.lane && !lane_ok)
            fail("node '" + n.decl->name + "': input '" + name + "' cannot take an array-valued ([f32; 32]) source");
        for (size_t i = 1; i < it->second.size(); ++i) { // connect, then accumulate in edge order
            Val b = cg.eval(it->second[i].e);
            if (acc.is_frame() || b.is_frame()) { // AccumulateEndpoints for Frame<C>: element-wise (static_context.rs:189-194)
                acc = cg.frame_arith(acc, b, '+', "fan-in of '" + name + "'");
                continue;
            }
            if (b.inner != dst_inner && b.rate == Rate::Vary)
                fail("fan-in summing supports only same-rate sources (input '" + name + "')");
            Val r;
            r.e = "(" + acc.e + " + " + b.e + ")";
            r.rate = join(acc.rate, b.rate);
            r.inner = acc.inner || b.inner;
            r.voice_inputs = acc.voice_inputs;
            r.voice_inputs.insert(b.voice_inputs.begin(), b.voice_inputs.end());
            if (acc.host && b.host && r.rate <= Rate::UBlock) {
                HostFn ha = acc.host, hb = b.host;
                r.host = [ha, hb](const UEnv& env) { return ha(env) + hb(env); };
            }
            acc = r;
        }
        return acc;
    }
    Val in_uniform(const std::string& name)
    {
        Val v = in(name);
        if (v.rate > Rate::UBlock || !v.host)
            fail_unsupported("node '" + n.decl->name + "': input '" + name +
                 "' must be a block-uniform value (constant or non-ramped broadcast input) in this version");
        return v;
    }
    std::string sf(int slot) const { return "SF(" + std::to_string(slot) + ")"; }
    std::string su(int slot) const { return "SU(" + std::to_string(slot) + ")"; }
    int slot_f(std::function<float(const UEnv&)> fn)
    {
        return cg.new_slot([fn](const UEnv& e) { return fbits(fn(e)); });
    }
    int slot_u(std::function<uint32_t(const UEnv&)> fn) { return cg.new_slot(std::move(fn)); }
    float rate_sr_factor() const { return (float)n.decl->rate_factor; }
    int sr_slot()
    {
        float k = rate_sr_factor();
        return slot_f([k](const UEnv& e) { return e.sample_rate * k; });
    }
    // declare a state word held in a register named <prefix><name>
    std::string state_f(const std::string& name, std::function<float(const UEnv&)> init)
    {
        std::string var = p + name;
        int w = cg.new_state(n.decl->name + "." + name, true, [init](const UEnv& e) { return fbits(init(e)); });
        cg.S().decl << "    float " << var << " = 0.0f;\n";
        cg.S().load << "        " << var << " = og::ld_f(A, c, " << w << ");\n";
        cg.S().store << "        og::st_f(A, c, " << w << ", " << var << ");\n";
        return var;
    }
    // an array field (`[f32; 32]`) of an LPV > 1 graph: every lane keeps OG_HPL of its elements in one og::HarmV.
    // store_if: the array is read-mostly -- written back only when that (per-lane) flag is set
    std::string state_lane_h(const std::string& name, float init, const std::string& store_if = std::string())
    {
        std::string var = p + name;
        cg.out.lane_state.push_back({n.decl->name + "." + name + "[h]", true, [init](const UEnv&) { return fbits(init); },
                                     !store_if.empty()});
        int k = (int)cg.out.lane_state.size() - 1;
        cg.S().decl << "    og::HarmV " << var << " = og::harm_splat(" << flit(init) << ");\n";
        cg.S().load << "        " << var << " = og::ldl_h<LPV>(A, c, " << k << ");\n";
        cg.S().store << "        " << (store_if.empty() ? std::string() : "if (" + store_if + ") ") << "og::stl_h<LPV>(A, c, " << k
                     << ", " << var << ");\n";
        return var;
    }
    // an array field that stays in its state plane (read-mostly: touched by event handlers, not by the frame loop): the
    // expression of this lane's pointer to its OG_HPL words
    std::string state_lane_plane(const std::string& name, float init)
    {
        cg.out.lane_state.push_back({n.decl->name + "." + name + "[h]", true, [init](const UEnv&) { return fbits(init); }, true});
        return "og::lane_plane<LPV>(A, c, " + std::to_string((int)cg.out.lane_state.size() - 1) + ")";
    }
    std::string state_u(const std::string& name, uint32_t init)
    {
        std::string var = p + name;
        int w = cg.new_state(n.decl->name + "." + name, false, [init](const UEnv&) { return init; });
        cg.S().decl << "    uint32_t " << var << " = 0u;\n";
        cg.S().load << "        " << var << " = og::ld_u(A, c, " << w << ");\n";
        cg.S().store << "        og::st_u(A, c, " << w << ", " << var << ");\n";
        return var;
    }
    void set_out(const std::string& port, const std::string& expr, bool patched_later = false)
    {
        std::string var = p + port;
        std::replace(var.begin(), var.end(), '#', '_');
        cg.os() << "        " << (patched_later ? "" : "const ") << "float " << var << " = " << expr << ";\n";
        Val v;
        v.e = var;
        v.rate = Rate::Vary;
        v.inner = n.domain == 1;
        const std::string key = "n" + std::to_string(n.id) + "." + port;
        cg.node_outputs[key] = v;
        auto fit = cg.fb_vars.find(key); // an earlier node of the frame reads last frame's value
        if (fit != cg.fb_vars.end()) cg.os() << "        " << fit->second << " = " << var << ";\n";
    }
    void set_out_frame(const std::string& port, const std::vector<std::string>& exprs)
    {
        for (size_t c = 0; c < exprs.size(); ++c) set_out(port + "#" + std::to_string(c), exprs[c]);
    }
    // a value that is constant over the block per voice: computed in derive()
    std::string hoist(const std::string& name, const std::string& expr)
    {
        std::string var = p + name;
        cg.S().decl << "    float " << var << " = 0.0f;\n";
        cg.S().derive << "        " << var << " = " << expr << ";\n";
        cg.any_derive = true;
        return var;
    }
};

// ---- emitters ---------------------------------------------------------------

constexpr float ADSR_MIN_TIME = 1.0e-5f;
constexpr float ADSR_CURVE_K = 4.6051702f;

void NodeCtx::on_event(const std::string& port, const std::function<std::string(const std::string&)>& gen)
{
    // `@OFF@` in the generated call = the EventInstance::frame_offset the handler receives: for an event of a GRAPH input the
    // offset inside its process_block call (times N for a node of the oversampled region: compute_event_rescale,
    // ir/lower.rs:846-852); for an event pushed by another NODE the offset the producer gave it, rescaled across a rate
    // boundary like the reference's drains (codegen/emit_edge.rs:86-99: outer -> inner saturating_mul(N), inner -> outer / N)
    auto with_off = [](std::string call, const std::string& off) {
        for (size_t pos = call.find("@OFF@"); pos != std::string::npos; pos = call.find("@OFF@", pos)) call.replace(pos, 5, off);
        return call;
    };
    auto ev = n.ev_edges.find(port);
    if (ev != n.ev_edges.end())
        for (int ei : ev->second)
            cg.S().ev_handlers[ei] << with_off(gen("ev.value"), n.domain == 1 ? "blk_off(f) * " + std::to_string(cg.N) + "u" : "blk_off(f)");
    auto nv = n.ev_node_edges.find(port);
    if (nv == n.ev_node_edges.end()) return;
    const NodeInst& src = cg.nodes[nv->second.first];
    // Across a rate boundary (round 4; the reference's drains, codegen/emit_frame.rs:341-374): outer -> inner events are
    // copied in front of the inner loop (step 5.5) and handled at step 6a of the SAME outer frame; inner -> outer events
    // -- everything the N inner ticks pushed -- are copied after the loop (step 7a) and handled when the outer node
    // runs (step 7.5).  An outer node that feeds the inner loop cannot read an inner node (it would have to run twice).
    if (src.domain != n.domain && !((src.domain == 0 && n.domain >= 1) || (src.domain == 1 && n.domain == 2)))
        fail("event edge '" + src.decl->name + "." + nv->second.second + " -> " + n.decl->name + "." + port +
             "' runs against the rate schedule (an outer node in front of the oversampled region cannot read a node of it)");
    if (cg.stage_of.size() > (size_t)n.id && cg.stage_of[n.id] != cg.stage_of[src.id])
        fail("internal: event edge across pipeline stages");
    const std::string q = "n" + std::to_string(src.id) + "_" + nv->second.second;
    // both ends in the oversampled region: the reference copies the producer's queue into the consumer's on every inner
    // tick but runs the consumer's handlers (process_event_inputs) once per OUTER tick, in front of the inner loop --
    // i.e. at the start of the NEXT outer frame, over everything the producer pushed during this one (step 6a,
    // codegen/emit_frame.rs:150-160; oscen-macros/src/lib.rs:266-285: process_event_inputs clears the node's own outputs,
    // then dispatches its input queues)
    const std::string Ns = std::to_string(cg.N) + "u";
    const std::string node_off = (src.domain == 0 && n.domain == 1) ? "og::sat_mul_u32(" + q + ".off(evk), " + Ns + ")"
                                 : (src.domain == 1 && n.domain == 2) ? "(" + q + ".off(evk) / " + Ns + ")"
                                                                      : q + ".off(evk)";
    const bool inner_pair = n.domain == 1 && src.domain <= 1; // the handler runs at step 6a, in front of the inner loop
    if (inner_pair && gen("evv").find("[j]") != std::string::npos)
        fail_unsupported("node '" + n.decl->name + "': a handler fed by another oversampled node runs in front of the inner loop; "
                         "its value inputs cannot come through a resampled edge in this version");
    (inner_pair ? cg.S().s_ev6a : cg.os())
            << "        if (__any((int)(" << q << ".n != 0u))) { // events '" << src.decl->name << "." << nv->second.second
            << (inner_pair && src.domain == 1 ? "' pushed during the previous outer frame\n" : "' pushed on this frame\n")
            << "            OG_EV_LOOP_PRAGMA\n" // (unrolled up to a capacity of 4, a rolled loop above: the handler is inlined per copy)
            << "            for (uint32_t evk = 0; evk < OG_NODE_EVENTS_PER_FRAME; ++evk)\n"
            << "                if (OG_NODE_EVENTS_PER_FRAME > 4 && !__any((int)(evk < " << q << ".n))) break;\n"
            << "                else if (evk < " << q << ".n) {\n"
            << "                const float evv = " << q << ".get(evk);\n"
            << with_off(gen("evv"), node_off) << "                }\n"
            << "        }\n";
}

uint32_t rs_as_u32(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
// adsr.rs:117-134 on the host: same libm the reference's f32::exp binds to
uint32_t adsr_samples(float t, float sr)
{
    uint32_t n = rs_as_u32(fmaxf(fmaxf(t, 0.0f), ADSR_MIN_TIME) * fmaxf(sr, 1.0f));
    return n < 1 ? 1 : n;
}
float adsr_coeff(uint32_t n) { return 1.0f - expf(-ADSR_CURVE_K / (float)n); }

void emit_adsr(NodeCtx& x)
{
    Val a = x.in_uniform("attack"), d = x.in_uniform("decay"), s = x.in_uniform("sustain"),
        r = x.in_uniform("release");
    const float k = x.rate_sr_factor();
    HostFn ha = a.host, hd = d.host, hs = s.host, hr = r.host;
    int s_an = x.slot_u([ha, k](const UEnv& e) { return adsr_samples(ha(e), e.sample_rate * k); });
    int s_dn = x.slot_u([hd, k](const UEnv& e) { return adsr_samples(hd(e), e.sample_rate * k); });
    int s_rn = x.slot_u([hr, k](const UEnv& e) { return adsr_samples(hr(e), e.sample_rate * k); });
    int s_ac = x.slot_f([ha, k](const UEnv& e) { return adsr_coeff(adsr_samples(ha(e), e.sample_rate * k)); });
    int s_dc = x.slot_f([hd, k](const UEnv& e) { return adsr_coeff(adsr_samples(hd(e), e.sample_rate * k)); });
    int s_su = x.slot_f([hs](const UEnv& e) {
        float v = hs(e);
        return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    });
    int s_ai = x.slot_u([ha](const UEnv& e) { return (uint32_t)(ha(e) <= ADSR_MIN_TIME); });
    int s_ri = x.slot_u([hr](const UEnv& e) { return (uint32_t)(fmaxf(hr(e), 0.0f) <= ADSR_MIN_TIME); });
    (void)s_rn; (void)s_su; (void)s_ai; (void)s_ri; // eight consecutive slots, see og_nodes.hip.h
    const std::string K = "A, " + std::to_string(s_an);
    const std::string E = x.p + "e";
    // state planes keep the reference's fields; the kernel works on the register form og::Adsr
    int w_stage = x.cg.new_state(x.n.decl->name + ".stage", false, [](const UEnv&) { return 0u; });
    int w_rem = x.cg.new_state(x.n.decl->name + ".samples_remaining", false, [](const UEnv&) { return 0u; });
    int w_level = x.cg.new_state(x.n.decl->name + ".level", true, [](const UEnv&) { return fbits(0.0f); });
    int w_vel = x.cg.new_state(x.n.decl->name + ".velocity", true, [](const UEnv&) { return fbits(1.0f); });
    x.cg.S().decl << "    og::Adsr " << E << " = {0u, og::ADSR_HOLD, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 4294967296.0f};\n";
    x.cg.S().load << "        og::adsr_block_begin(" << E << ", og::ld_u(A, c, " << w_stage << "), og::ld_u(A, c, " << w_rem
              << "), og::ld_f(A, c, " << w_level << "), og::ld_f(A, c, " << w_vel << "), " << K << ");\n";
    x.cg.S().store << "        og::st_u(A, c, " << w_stage << ", " << E << ".stage);\n"
               << "        og::st_u(A, c, " << w_rem << ", og::adsr_rem(" << E << "));\n"
               << "        og::st_f(A, c, " << w_level << ", " << E << ".lv);\n"
               << "        og::st_f(A, c, " << w_vel << ", " << E << ".vel);\n";
    x.on_event("gate", [&](const std::string& val) { return "                og::adsr_gate(" + E + ", " + val + ", " + K + ");\n"; });
    // (outer-rate envelopes: the chunk loops keep <E>.fc == (float)<E>.cnt at their top, see og::Adsr::fc and fc_sync below)
    x.set_out("output", x.n.domain == 1 ? "og::adsr_tick<true, false>(" + E + ")" : "og::adsr_tick<decltype(chk)::release, true>(" + E + ")", true);
    const std::string fix = "og::adsr_complete(" + E + ", " + x.sf(s_ac) + ", " + x.sf(s_dc) + ", " + x.su(s_dn) + ", " + x.p +
                            "output);\n";
    if (x.n.domain == 1) // oversampled: N ticks per frame, finish a stage end right away
        x.cg.os() << "        " << fix;
    else // one check for a whole run of envelopes, emitted before anything can read their outputs
        x.cg.pending_post.push_back({E + ".cnt", "            " + fix});
}

void emit_fm_operator(NodeCtx& x)
{
    Val bf = x.in("base_freq"), ratio = x.in("ratio"), pm = x.in("phase_mod"), fb = x.in("feedback"),
        env = x.in("envelope"), lvl = x.in("level");
    int s_sr = x.sr_slot();
    std::string phase = x.state_f("phase", [](const UEnv&) { return 0.0f; });
    std::string prev = x.state_f("prev_output", [](const UEnv&) { return 0.0f; });
    std::string inc_expr = "(" + bf.e + " * " + ratio.e + ") / " + x.sf(s_sr);
    std::string inc;
    Rate rr = join(bf.rate, ratio.rate);
    if (rr <= Rate::VBlock && rr != Rate::UFrame) {
        inc = x.hoist("inc", inc_expr);
    } else {
        inc = x.p + "inc";
        x.cg.os() << "        const float " << inc << " = " << inc_expr << ";\n";
    }
    // unconnected feedback (0.0) contributes prev*0 + pm = pm for every finite prev: drop the two ops
    const bool no_fb = fb.rate == Rate::Const && !x.connected("feedback") && x.def("feedback") == 0.0f;
    x.set_out("output", std::string(no_fb ? "og::fm_operator_tick_nofb(" : "og::fm_operator_tick(") + phase + ", " + prev +
                            ", " + inc + ", " + pm.e + ", " + (no_fb ? "" : fb.e + ", ") + env.e + ", " + lvl.e + ")");
}

void emit_tpt(NodeCtx& x)
{
    // TptFilter<F: AudioFrame> (tpt/mod.rs:104-123): scalar coefficients, one pair of integrators per channel
    const int width = x.n.type->out_channels.empty() ? 1 : x.n.type->out_channels[0];
    Val in = width > 1 ? x.in_frame("input", width) : x.in("input");
    Val cutoff = x.in("cutoff"), q = x.in("q"), fmod = x.in("f_mod");
    const float k = x.rate_sr_factor();
    const float c0 = x.def("cutoff"), q0 = x.def("q");
    int s_two_sr = x.slot_f([k](const UEnv& e) { return 2.0f * (e.sample_rate * k); });
    int s_period = x.slot_f([k](const UEnv& e) { return 0.5f / (e.sample_rate * k); });
    int s_nyq = x.slot_f([k](const UEnv& e) { return (e.sample_rate * k) * 0.5f - 1.1920929e-7f; });
    int s_maxc = x.slot_f([k](const UEnv& e) { return fminf((e.sample_rate * k) * 0.5f - 1.1920929e-7f, 20000.0f); });
    // prepare(): update_coefficients(sr, self.cutoff, self.q) with the ctor fields (tpt/mod.rs:129-131, 69-82)
    auto coef = [c0, q0, k](const UEnv& e, int which) {
        const float sr = e.sample_rate * k;
        const float nyquist = sr * 0.5f - 1.1920929e-7f;
        float freq = c0 < 20.0f ? 20.0f : c0;
        freq = freq > nyquist ? nyquist : freq;
        const float period = 0.5f / sr;
        const float f = (2.0f * sr) * tanf(2.0f * 3.14159274101257324f * freq * period) * period;
        const float inv_q = 1.0f / q0;
        const float h = 1.0f / (1.0f + inv_q * f + f * f);
        return which == 0 ? h : (which == 1 ? f : f + inv_q);
    };
    std::vector<std::string> z0s, z1s;
    for (int c = 0; c < width; ++c) {
        const std::string sfx = width > 1 ? "_" + std::to_string(c) : "";
        z0s.push_back(x.state_f("z0" + sfx, [](const UEnv&) { return 0.0f; }));
        z1s.push_back(x.state_f("z1" + sfx, [](const UEnv&) { return 0.0f; }));
    }
    const std::string z0 = z0s[0], z1 = z1s[0];
    std::string cc = x.state_f("current_cutoff", [c0](const UEnv&) { return c0; });
    std::string cq = x.state_f("current_q", [q0](const UEnv&) { return q0; });
    std::string h = x.state_f("h", [coef](const UEnv& e) { return coef(e, 0); });
    std::string g = x.state_f("g", [coef](const UEnv& e) { return coef(e, 1); });
    std::string kk = x.state_f("k", [coef](const UEnv& e) { return coef(e, 2); });
    const bool nomod = (fmod.rate == Rate::Const) && !x.connected("f_mod") && x.def("f_mod") == 0.0f;
    auto block_const = [](const Val& v) { return v.rate <= Rate::VBlock && v.rate != Rate::UFrame; };
    const std::string tail = x.sf(s_maxc) + ", " + x.sf(s_two_sr) + ", " + x.sf(s_period) + ", " + x.sf(s_nyq) + ", " + cc + ", " +
                             cq + ", " + h + ", " + g + ", " + kk + ");\n";
    if (nomod && block_const(cutoff) && block_const(q) && x.n.domain != 1) {
        // cutoff and q cannot change between events: apply_parameter_updates() finds nothing to do after the
        // first frame of the block, so it runs in derive() (block start and after per-voice value events)
        x.cg.S().derive << "        og::tpt_params_nomod(" << cutoff.e << ", " << q.e << ", " << tail;
        x.cg.any_derive = true;
    } else if (nomod && ogabi::experiment_knob("OGC_TPT_FLAT") && atoi(ogabi::experiment_knob("OGC_TPT_FLAT")) != 0) {
        // experiment: branch-free update (see og::tpt_params_nomod_flat); measured slower on MI355X for
        // fm_voice (65 536 voices 0.098 ms against 0.087 ms; 262 144 voices 0.293 against 0.234): the
        // branch is wave-uniformly skipped whenever no lane's envelope moves

        std::string iq;
        if (block_const(q)) iq = x.hoist("inv_q", "1.0f / og::clampf(" + q.e + ", 0.1f, 10.0f)");
        else iq = "(1.0f / og::clampf(" + q.e + ", 0.1f, 10.0f))";
        x.cg.os() << "        og::tpt_params_nomod_flat(" << cutoff.e << ", " << q.e << ", " << iq << ", " << tail;
    } else if (nomod && x.n.domain != 1 && !(ogabi::experiment_knob("OGC_TPT_LAZY") && atoi(ogabi::experiment_knob("OGC_TPT_LAZY")) == 0)) {
        // cutoff computed per frame: watch the raw input, run the reference's test only on frames whose input differs from
        // the previous frame's (og::tpt_params_nomod_lazy).  q is watched too when it can change inside a launch: a ramped
        // input (the RAMPS variants of the kernel) or a per-frame value.
        const std::string li = x.p + "last_in", lq = x.p + "last_q";
        x.cg.S().decl << "    float " << li << " = __uint_as_float(og::TPT_LAZY_SENTINEL), " << lq << " = __uint_as_float(og::TPT_LAZY_SENTINEL);\n";
        // a per-voice value event may change a block-constant q: start over (the event path runs derive())
        x.cg.S().derive << "        " << li << " = __uint_as_float(og::TPT_LAZY_SENTINEL);\n";
        x.cg.any_derive = true;
        const std::string qchk = block_const(q) ? "false" : (q.rate == Rate::UFrame ? "RAMPS" : "true");
        // 1 / q once per launch where q cannot change inside it: a block-constant q, or a rampable graph input in the kernel
        // variants that do not read the ramp table (`RV(row, slot)` is the slot there)
        std::string q_launch = block_const(q) ? q.e : std::string();
        if (q_launch.empty() && q.rate == Rate::UFrame && q.e.compare(0, 3, "RV(") == 0 && q.e.find(')') == q.e.size() - 1) {
            const size_t comma = q.e.find(", ");
            if (comma != std::string::npos) q_launch = "SF(" + q.e.substr(comma + 2);
        }
        // (a voice-uniform q -- a graph input that is not per-voice -- gives a wave-uniform quotient: kept in a scalar register)
        const bool q_uniform = q.rate <= Rate::UBlock || q.rate == Rate::UFrame;
        const std::string iq_expr = "1.0f / og::clampf(" + q_launch + ", 0.1f, 10.0f)";
        const std::string iq = q_launch.empty() ? std::string("0.0f") : x.hoist("inv_q", q_uniform ? "og::uniform_f(" + iq_expr + ")" : iq_expr);
        x.cg.os() << "        og::tpt_params_nomod_lazy<" << qchk << ", " << (q_launch.empty() ? "false" : "true") << ">(" << cutoff.e << ", " << q.e << ", "
                  << iq << ", " << li << ", " << lq << ", " << tail;
    } else if (nomod) {
        x.cg.os() << "        og::tpt_params_nomod(" << cutoff.e << ", " << q.e << ", " << tail;
    } else {
        x.cg.os() << "        og::tpt_params_mod(" << cutoff.e << ", " << q.e << ", " << fmod.e << ", " << tail;
    }
    if (width == 1) {
        x.set_out("output", "og::tpt_tick(" + in.e + ", " + z0 + ", " + z1 + ", " + h + ", " + g + ", " + kk + ")");
        return;
    }
    std::vector<std::string> outs;
    for (int c = 0; c < width; ++c)
        outs.push_back("og::tpt_tick(" + in.ch[c].e + ", " + z0s[c] + ", " + z1s[c] + ", " + h + ", " + g + ", " + kk + ")");
    x.set_out_frame("output", outs);
}

void emit_polyblep(NodeCtx& x)
{
    Val pm = x.in("phase_mod"), fr = x.in("frequency"), fm = x.in("frequency_mod"), amp = x.in("amplitude"),
        pw = x.in("pulse_width");
    int s_sr = x.sr_slot();
    std::string phase = x.state_f("phase", [](const UEnv&) { return 0.0f; });
    // frequency * (1 + frequency_mod) and its per-sample increment (an IEEE division): per block for a
    // voice whose frequency inputs only change through events, per tick otherwise
    const std::string f_expr = "og::polyblep_frequency(" + fr.e + ", " + fm.e + ")";
    std::string f, inc;
    const Rate rr = join(fr.rate, fm.rate);
    if (rr <= Rate::VBlock && rr != Rate::UFrame) {
        f = x.hoist("freq", f_expr);
        inc = x.hoist("inc", "og::polyblep_increment(" + f + ", " + x.sf(s_sr) + ")");
    } else {
        f = x.p + "freq";
        inc = x.p + "inc";
        x.cg.os() << "        const float " << f << " = " << f_expr << ";\n"
                  << "        const float " << inc << " = og::polyblep_increment(" << f << ", " << x.sf(s_sr) << ");\n";
    }
    // the >= sr/4 sine fallback is a per-voice block constant when the frequency is: a chunk variant drops it
    const bool steady_ok = rr <= Rate::VBlock && rr != Rate::UFrame && x.n.type->variant != 0;
    if (steady_ok) x.cg.S().fast_conds.push_back("(" + f + " < " + x.sf(s_sr) + " * 0.25f)");
    x.set_out("output", "og::polyblep_tick<" + std::to_string(x.n.type->variant) + "u" +
                            (steady_ok ? ", decltype(chk)::steady" : "") + ">(" + phase + ", " + f + ", " + inc + ", " + pm.e +
                            ", " + amp.e + ", " + pw.e + ", " + x.sf(s_sr) + ")");
}

void emit_oscillator(NodeCtx& x)
{
    Val fr = x.in("frequency"), fm = x.in("frequency_mod"), amp = x.in("amplitude");
    int s_sr = x.sr_slot();
    std::string phase = x.state_f("phase", [](const UEnv&) { return 0.0f; });
    x.set_out("output", "og::oscillator_tick<" + std::to_string(x.n.type->variant) + "u>(" + phase + ", " + fr.e +
                            ", " + fm.e + ", " + amp.e + ", " + x.sf(s_sr) + ")");
}

// IirLowpass (filters/iir_lowpass/mod.rs): coefficients are refreshed when frame_counter == 0 (every
// 32nd tick).  With block-uniform cutoff/q the five coefficients are host-derived slots (the host's
// libm tanf is the one the reference's f32::tan binds to); otherwise they are computed on the device.
struct IirCoef {
    float b0, b1, b2, a1, a2;
};
IirCoef iir_lowpass_coef(float cutoff, float q_in, float sr)
{
    const float nyquist = sr * 0.5f - 1.1920929e-7f;
    float freq = cutoff < 20.0f ? 20.0f : cutoff;
    freq = freq > nyquist ? nyquist : freq;
    const float q = fmaxf(q_in, 0.01f);
    const float n = 1.0f / tanf(3.14159274101257324f * freq / sr);
    const float n2 = n * n;
    const float c1 = 1.0f / (1.0f + 1.0f / q * n + n2);
    return {c1, c1 * 2.0f, c1, c1 * 2.0f * (1.0f - n2), c1 * (1.0f - 1.0f / q * n + n2)};
}

void emit_iir_lowpass(NodeCtx& x)
{
    const Val in = x.in("input");
    const Val cutoff = x.in("cutoff");
    const Val q = x.in("q");
    const float k = x.rate_sr_factor();
    const float c0 = x.def("cutoff"), q0 = x.def("q");
    auto prep = [c0, q0, k](int which) {
        return [c0, q0, k, which](const UEnv& e) {
            const IirCoef c = iir_lowpass_coef(c0, q0, e.sample_rate * k);
            const float v[5] = {c.b0, c.b1, c.b2, c.a1, c.a2};
            return v[which];
        };
    };
    std::string v1 = x.state_f("v1", [](const UEnv&) { return 0.0f; });
    std::string v2 = x.state_f("v2", [](const UEnv&) { return 0.0f; });
    std::string b0 = x.state_f("b0", prep(0)), b1 = x.state_f("b1", prep(1)), b2 = x.state_f("b2", prep(2)),
                a1 = x.state_f("a1", prep(3)), a2 = x.state_f("a2", prep(4));
    std::string fc = x.state_u("frame_counter", 0);
    if (cutoff.rate <= Rate::UBlock && q.rate <= Rate::UBlock && cutoff.host && q.host) {
        HostFn hc = cutoff.host, hq = q.host;
        int s[5];
        for (int w = 0; w < 5; ++w)
            s[w] = x.slot_f([hc, hq, k, w](const UEnv& e) {
                const IirCoef c = iir_lowpass_coef(hc(e), hq(e), e.sample_rate * k);
                const float v[5] = {c.b0, c.b1, c.b2, c.a1, c.a2};
                return v[w];
            });
        x.cg.os() << "        if (" << fc << " == 0u) { " << b0 << " = " << x.sf(s[0]) << "; " << b1 << " = " << x.sf(s[1]) << "; "
                  << b2 << " = " << x.sf(s[2]) << "; " << a1 << " = " << x.sf(s[3]) << "; " << a2 << " = " << x.sf(s[4])
                  << "; }\n";
    } else {
        int s_sr = x.sr_slot();
        int s_nyq = x.slot_f([k](const UEnv& e) { return (e.sample_rate * k) * 0.5f - 1.1920929e-7f; });
        x.cg.os() << "        if (" << fc << " == 0u) og::iir_lowpass_coeffs(" << cutoff.e << ", " << q.e << ", " << x.sf(s_sr)
                  << ", " << x.sf(s_nyq) << ", " << b0 << ", " << b1 << ", " << b2 << ", " << a1 << ", " << a2 << ");\n";
    }
    x.cg.os() << "        " << fc << " = (" << fc << " + 1u) & 31u;\n";
    x.set_out("output", "og::iir_lowpass_tick(" + in.e + ", " + v1 + ", " + v2 + ", " + b0 + ", " + b1 + ", " + b2 + ", " +
                            a1 + ", " + a2 + ")");
}

// LP18Filter (examples/nih-twin-peaks/src/lp18_filter.rs)
void emit_lp18(NodeCtx& x)
{
    const Val in = x.in("input");
    const Val cutoff = x.in("cutoff");
    const Val fmod = x.in("fmod");
    Val res = x.in("resonance");
    const float k = x.rate_sr_factor();
    const float c0 = x.def("cutoff"), r0 = x.def("resonance");
    auto clamp_res = [](float r) { return r < 0.0f ? 0.0f : (r > 0.99f ? 0.99f : r); };
    if (!x.connected("resonance")) res = vconst(clamp_res(r0)); // new() stores the clamped value in the field :50
    int s_sr = x.sr_slot();
    std::string z0 = x.state_f("z0", [](const UEnv&) { return 0.0f; });
    std::string z1 = x.state_f("z1", [](const UEnv&) { return 0.0f; });
    std::string z2 = x.state_f("z2", [](const UEnv&) { return 0.0f; });
    std::string g = x.state_f("g", [c0, k](const UEnv& e) { // prepare() :75-78
        float fc = (c0 + 0.0f) / (e.sample_rate * k);
        fc = fc < 0.001f ? 0.001f : (fc > 0.33f ? 0.33f : fc);
        return tanf(3.14159274101257324f * fc);
    });
    std::string h = x.state_f("h", [r0, clamp_res](const UEnv&) { return 2.0f * clamp_res(r0); });
    std::string lc = x.state_f("last_cutoff", [c0](const UEnv&) { return c0; });
    std::string lf = x.state_f("last_fmod", [](const UEnv&) { return 0.0f; });
    std::string lr = x.state_f("last_resonance", [r0](const UEnv&) { return r0; });
    x.cg.os() << "        og::lp18_params(" << cutoff.e << ", " << fmod.e << ", " << res.e << ", " << x.sf(s_sr) << ", " << g
              << ", " << h << ", " << lc << ", " << lf << ", " << lr << ");\n";
    x.set_out("output", "og::lp18_tick(" + in.e + ", " + z0 + ", " + z1 + ", " + z2 + ", " + g + ", " + h + ")");
}

// Delay (oscen-lib/src/delay/mod.rs): the line itself is an HBM ring per voice (CompiledGraph::rings)
void emit_delay(NodeCtx& x)
{
    // (array-valued graphs: the line stays one ring per VOICE; the lanes of a voice hold the same per-voice scalars, read
    //  the same slots and write the same value to the same slot -- each lane's loads see its own stores)
    if (x.cg.out.rings.size() >= 4) fail("at most 4 Delay nodes per graph");
    const Val in = x.in("input");
    const Val ds = x.in("delay_samples");
    const Val fb = x.in("feedback");
    const int k = (int)x.cg.out.rings.size();
    x.cg.out.rings.push_back({x.n.decl->name, x.rate_sr_factor()});
    const float d0 = x.def("delay_samples"), f0 = x.def("feedback");
    std::string dsv = x.state_f("delay_samples", [d0](const UEnv&) { return d0; });
    std::string fbv = x.state_f("feedback", [f0](const UEnv&) { return f0; });
    std::string wp = x.state_u("write_pos", 0), fc = x.state_u("frame_counter", 0);
    // connected value inputs overwrite the field every frame; the clamp of frame_counter == 0 then applies
    // to that frame's value only (delay/mod.rs:47-56 after the generated edge copy)
    if (x.connected("delay_samples")) x.cg.os() << "        " << dsv << " = " << ds.e << ";\n";
    if (x.connected("feedback")) x.cg.os() << "        " << fbv << " = " << fb.e << ";\n";
    // whole-sample reads are staged a chunk ahead (og::ring_chunk_begin); the offset the staging assumes is
    // the connected input when that is constant over the block, else the field as the last tick left it
    const std::string K = std::to_string(k), pre = x.p + "pre";
    const bool hint_input = x.connected("delay_samples") && ds.rate <= Rate::VBlock && ds.rate != Rate::UFrame;
    auto block_const = [](const Val& v) { return v.rate <= Rate::VBlock && v.rate != Rate::UFrame; };
    const bool fixed = (!x.connected("delay_samples") || block_const(ds)) && (!x.connected("feedback") || block_const(fb));
    x.cg.S().decl << "    og::RingPre " << pre << " = {og::RING_NONE, og::RING_NONE, false, {}};\n";
    if (x.n.domain == 1) {
        // `Delay::new(..) * N`: the line runs at the oversampled rate (its ring is sized from sr * N, RingSpec::capacity), N
        // ticks per outer frame.  The chunk-ahead staging assumes one tick per frame; here every tick reads its sample
        // directly (the RingPre above stays empty, so og::delay_tick takes its un-staged path).
        x.set_out("output", "og::delay_tick(A.rings[" + K + "], A.ring_cap[" + K + "], A.n_voices, c.v, c.valid, " + in.e + ", " +
                                dsv + ", " + fbv + ", " + wp + ", " + fc + ", " + pre + ", ring_lds[" + K + "], c.lane, 0u)");
        return;
    }
    x.cg.S().chunk_begin << "        og::ring_chunk_begin(A.rings[" << K << "], A.ring_cap[" << K << "], A.n_voices, c.v, c.valid, "
                         << (hint_input ? ds.e : dsv) << ", " << ((x.connected("feedback") && block_const(fb)) ? fb.e : fbv)
                         << ", " << (fixed ? "true" : "false") << ", " << wp << ", " << pre << ", ring_lds[" << K
                         << "], c.lane, base + OG_BUS_CHUNK < A.frames);\n";
    x.set_out("output", "og::delay_tick(A.rings[" + K + "], A.ring_cap[" + K + "], A.n_voices, c.v, c.valid, " + in.e + ", " +
                            dsv + ", " + fbv + ", " + wp + ", " + fc + ", " + pre + ", ring_lds[" + K + "], c.lane, f - cbase)");
}

// (inputs are resolved in separate statements: operand evaluation order is unspecified in C++ and
//  resolving an input can allocate cut-crossing channels, whose numbering must be deterministic)
void emit_binary(NodeCtx& x, const char* a, const char* b, const char* op)
{
    const Val va = x.in(a);
    const Val vb = x.in(b);
    x.set_out("output", va.e + op + vb.e);
}
void emit_gain(NodeCtx& x) { emit_binary(x, "input", "gain", " * "); }
void emit_vca(NodeCtx& x) { emit_binary(x, "input", "control", " * "); }
void emit_add_value(NodeCtx& x) { emit_binary(x, "input", "value", " + "); }
void emit_mixer(NodeCtx& x) { emit_binary(x, "input_a", "input_b", " + "); }
void emit_hardclip(NodeCtx& x) { x.set_out("output", "og::hardclip(" + x.in("input").e + ")"); }
void emit_crossfade(NodeCtx& x)
{
    Val in = x.in("input"), mix = x.in("mix");
    std::string m = x.p + "mix";
    x.cg.os() << "        const float " << m << " = og::clamp01(" << mix.e << ");\n";
    x.set_out("output_a", in.e + " * (1.0f - " + m + ")");
    x.set_out("output_b", in.e + " * " + m);
}

// electric piano (examples/electric-piano/src/electric_piano_voice.rs), one voice = 8 lanes x 4 harmonics
void emit_ep_amp(NodeCtx& x)
{
    Val br = x.in("brightness"), vs = x.in("velocity_scaling"), dr = x.in("decay_rate"), hd = x.in("harmonic_decay"),
        ks = x.in("key_scaling"), rr = x.in("release_rate");
    (void)x.in("frequency"); // routed to the node but never read by its process()
    const std::string A = x.p + "a";
    // decay / release only change when a note starts and are read only by the gate handler: they stay in their state planes
    // (og_nodes.hip.h, EpAmp); the table the next ramp target uses sits in an LDS column
    std::string cur = x.state_lane_h("current_value", 0.0f), tgt = x.state_lane_h("target_value", 0.0f);
    const std::string dec = x.state_lane_plane("decay", 0.0f), rel = x.state_lane_plane("release", 0.0f);
    auto writable = [](std::string e) { return e.replace(e.find("og::lane_plane<"), 15, "og::lane_plane_or_dump<"); }; // (handlers: see there)
    const std::string dec_w = writable(dec), rel_w = writable(rel);
    std::string released = x.state_u("released", 0), step = x.state_u("interpolation_step", 64);
    std::string vel = x.state_f("velocity", [](const UEnv&) { return 0.0f; });
    x.cg.S().decl << "    og::EpAmp " << A << " = {};\n";
    x.cg.S().load << "        " << A << " = og::EpAmp{" << cur << ", " << tgt << ", " << released << ", " << step << ", " << vel
              << "};\n";
    x.cg.S().pre << "    og::ep_amp_begin(" << A << ", " << dec << ", " << rel << ", c.valid);\n";
    // stores run before the generic store section reads the mirrors back
    x.cg.S().pre_store << "        " << cur << " = " << A << ".cur; " << tgt << " = " << A << ".tgt; " << released << " = " << A
                   << ".released; " << step << " = " << A << ".step; " << vel << " = " << A << ".velocity;\n";
    x.on_event("gate", [&](const std::string& val) {
        return "                og::ep_amp_gate(" + A + ", " + dec_w + ", " + rel_w + ", c.h * OG_HPL, " + val + ", " + br.e + ", " + vs.e + ", " + dr.e + ", " + hd.e +
               ", " + ks.e + ", " + rr.e + ");\n";
    });
    const std::string var = x.p + "amplitudes";
    x.cg.os() << "        const og::HarmV " << var << " = og::ep_amp_tick(" << A << ");\n";
    Val v;
    v.e = var;
    v.rate = Rate::Vary;
    v.lane = true;
    x.cg.node_outputs["n" + std::to_string(x.n.id) + ".amplitudes"] = v;
}

void emit_ep_bank(NodeCtx& x)
{
    Val fr = x.in("frequency"), amp = x.in_lane("amplitudes");
    if (!amp.lane) fail("node '" + x.n.decl->name + "': 'amplitudes' needs an array-valued source (AmplitudeSource.amplitudes)");
    int s_sr = x.sr_slot();
    const std::string B = x.p + "b";
    // the rotation multipliers only change with the frequency: read-mostly planes
    std::string re = x.state_lane_h("osc_re", 1.0f), im = x.state_lane_h("osc_im", 0.0f),
                mre = x.state_lane_h("mul_re", 1.0f, B + ".mul_dirty"), mim = x.state_lane_h("mul_im", 0.0f, B + ".mul_dirty");
    std::string lf = x.state_f("last_frequency", [](const UEnv&) { return 0.0f; });
    x.cg.S().decl << "    og::EpBank " << B << " = {};\n";
    x.cg.S().load << "        " << B << " = og::EpBank{" << re << ", " << im << ", " << mre << ", " << mim << ", " << lf << ", false};\n";
    x.cg.S().pre_store << "        " << re << " = " << B << ".re; " << im << " = " << B << ".im; " << mre << " = " << B << ".mre; "
                   << mim << " = " << B << ".mim; " << lf << " = " << B << ".last_frequency;\n";
    x.on_event("gate", [&](const std::string& val) { return "                og::ep_bank_gate(" + B + ", " + val + ");\n"; });
    // update_multipliers(): the frequency test of process() can only fire when the frequency changed
    const std::string upd = "og::ep_bank_update(" + B + ", c.h * OG_HPL, " + fr.e + ", " + x.sf(s_sr) + ");\n";
    if (fr.rate <= Rate::VBlock && fr.rate != Rate::UFrame) {
        x.cg.S().derive << "        " << upd;
        x.cg.any_derive = true;
    } else {
        x.cg.os() << "        " << upd;
    }
    // Does the output feed nothing but the graph output?  Then every lane may hand its own share to the
    // mix bus instead of folding the voice's lanes first (taps still need the folded voice output).
    int uses = 0;
    bool only_bus = true;
    const std::string me = x.n.decl->name + ".output";
    for (const GEdge& e : x.cg.g.edges) {
        if (e.src.find(x.n.decl->name + ".") == std::string::npos) continue;
        std::string t = e.src;
        t.erase(std::remove_if(t.begin(), t.end(), [](char ch) { return isspace((unsigned char)ch) || ch == '(' || ch == ')'; }), t.end());
        ++uses;
        bool bus_dst = false;
        for (const GNode& nd : x.cg.g.nodes)
            if (nd.bus && e.dst.rfind(nd.name + ".", 0) == 0) bus_dst = true; // (post-mix nodes read the summed bus)
        if (t != me || (e.dst.find('.') != std::string::npos && !bus_dst) || !e.policy.empty()) only_bus = false;
    }
    const bool share = uses == 1 && only_bus;
    if (share) x.cg.bus_all_lanes = true;
    x.set_out("output", std::string("og::ep_bank_tick<") + (share ? "TAPS" : "true") + ">(" + B + ", " + amp.e + ")");
}

const std::map<std::string, NodeTypeInfo>& registry()
{
    static const std::map<std::string, NodeTypeInfo> R = [] {
        std::map<std::string, NodeTypeInfo> r;
        const Kind V = Kind::Value, S = Kind::Stream, E = Kind::Event;
        r["AdsrEnvelope::new"] = {{{"gate", E, 0, -1}, {"attack", V, 0, 0}, {"decay", V, 0, 1}, {"sustain", V, 0, 2},
                                   {"release", V, 0, 3}},
                                  {"output"}, emit_adsr, 0, 4};
        r["FmOperator::new"] = {{{"base_freq", V, 440.0f, -1}, {"ratio", V, 1.0f, -1}, {"phase_mod", S, 0, -1},
                                 {"feedback", V, 0, -1}, {"envelope", S, 1.0f, -1}, {"level", V, 1.0f, -1}},
                                {"output"}, emit_fm_operator, 0, 0};
        r["TptFilter::new"] = {{{"input", S, 0, -1}, {"cutoff", S, 0, 0}, {"q", V, 0, 1}, {"f_mod", S, 0, -1}},
                               {"output"}, emit_tpt, 0, 2};
        for (int w : {2, 4}) { // TptFilter::<Frame<N>>::new (Stereo / Quad)
            NodeTypeInfo t = r["TptFilter::new"];
            t.inputs[0].channels = w;
            t.out_channels = {w};
            r["TptFilter<" + std::to_string(w) + ">::new"] = t;
        }
        const char* pbn[4] = {"sine", "saw", "square", "triangle"};
        for (int w = 0; w < 4; ++w)
            r[std::string("PolyBlepOscillator::") + pbn[w]] = {
                {{"phase_mod", S, 0, -1}, {"frequency", V, 0, 0}, {"frequency_mod", S, 0, -1}, {"amplitude", V, 0, 1},
                 {"pulse_width", V, 0.5f, -1}},
                {"output"}, emit_polyblep, w, 2};
        const char* on[3] = {"sine", "square", "saw"};
        for (int w = 0; w < 3; ++w)
            r[std::string("Oscillator::") + on[w]] = {
                {{"frequency", V, 0, 0}, {"frequency_mod", S, 0, -1}, {"amplitude", V, 0, 1}},
                {"output"}, emit_oscillator, w, 2};
        r["Gain::new"] = {{{"input", S, 0, -1}, {"gain", S, 1.0f, 0}}, {"output"}, emit_gain, 0, 1};
        r["Vca::new"] = {{{"input", S, 0, -1}, {"control", S, 1.0f, -1}}, {"output"}, emit_vca, 0, 0};
        r["AddValue::new"] = {{{"input", S, 0, -1}, {"value", V, 0, 0}}, {"output"}, emit_add_value, 0, 1};
        r["Mixer::new"] = {{{"input_a", S, 0, -1}, {"input_b", S, 0, -1}}, {"output"}, emit_mixer, 0, 0};
        r["Crossfade::new"] = {{{"input", S, 0, -1}, {"mix", V, 0, -1}}, {"output_a", "output_b"}, emit_crossfade, 0, 0};
        r["IirLowpass::new"] = {{{"input", S, 0, -1}, {"cutoff", V, 0, 0}, {"q", V, 0, 1}}, {"output"}, emit_iir_lowpass, 0, 2};
        r["LP18Filter::new"] = {{{"input", S, 0, -1}, {"cutoff", V, 0, 0}, {"fmod", V, 0, -1}, {"resonance", V, 0, 1}},
                                {"output"}, emit_lp18, 0, 2};
        r["Delay::new"] = {{{"input", S, 0, -1}, {"delay_samples", V, 0, 0}, {"feedback", V, 0, 1}}, {"output"}, emit_delay, 0, 2};
        r["HardClip::new"] = {{{"input", S, 0, -1}}, {"output"}, emit_hardclip, 0, 0};
        r["AmplitudeSource::new"] = {{{"frequency", V, 440.0f, -1}, {"gate", E, 0, -1}, {"brightness", V, 30.0f, -1},
                                      {"velocity_scaling", V, 50.0f, -1}, {"decay_rate", V, 90.0f, -1},
                                      {"harmonic_decay", V, 70.0f, -1}, {"key_scaling", V, 50.0f, -1},
                                      {"release_rate", V, 40.0f, -1}},
                                     {"amplitudes"}, emit_ep_amp, 0, 0, 8};
        r["OscillatorBank::new"] = {{{"frequency", V, 440.0f, -1}, {"gate", E, 0, -1}, {"amplitudes", S, 0, -1}},
                                    {"output"}, emit_ep_bank, 0, 0, 8};
        // post-mix only (og_graph_add_bus_node): examples/electric-piano/src/tremolo.rs
        r["Tremolo::new"] = {{{"input", S, 0, -1}, {"rate", V, 5.0f, -1}, {"depth", V, 0.5f, -1}}, {"output"}, nullptr, 0, 0};
        return r;
    }();
    return R;
}


// ---- user node types (og_register_node): the #[derive(Node)] plug-in surface ---------------------------
struct UserEntry {
    UserNodeType t;
    NodeTypeInfo info;
};
std::map<std::string, std::unique_ptr<UserEntry>>& user_registry()
{
    static std::map<std::string, std::unique_ptr<UserEntry>> R;
    return R;
}

std::string sanitize(const std::string& s)
{
    std::string r;
    for (char c : s) r.push_back(isalnum((unsigned char)c) ? c : '_');
    return r;
}

bool is_ident(const std::string& s)
{
    if (s.empty() || !(isalpha((unsigned char)s[0]) || s[0] == '_')) return false;
    for (char c : s)
        if (!(isalnum((unsigned char)c) || c == '_')) return false;
    return true;
}

// one tick of a user node: `og_user_<Type>_process(inputs..., state&..., outputs&..., sample_rate)`
void emit_user(NodeCtx& x)
{
    const UserNodeType& u = *x.n.type->user;
    const std::string fn = "og_user_" + sanitize(u.type);
    std::vector<Val> ins;
    std::vector<const UserPort*> in_ports;
    for (const UserPort& p : u.inputs) {
        if (p.kind == Kind::Event) continue;
        // (one statement per input: resolving an input can allocate cut-crossing channels, whose numbering must be deterministic)
        Val v = p.channels > 1 ? x.in_frame(p.name, p.channels) : x.in(p.name);
        if (v.is_frame()) { // handed over as one og::Frame<N> object
            std::string init;
            for (const Val& c : v.ch) init += (init.empty() ? "" : ", ") + c.e;
            v.e = "og::Frame<" + std::to_string(p.channels) + ">{{" + init + "}}";
        }
        ins.push_back(v);
        in_ports.push_back(&p);
    }
    const int s_sr = x.sr_slot();
    std::vector<std::string> st;
    for (const UserState& f : u.state) {
        if (f.is_uint) {
            st.push_back(x.state_u(f.name, f.init_u));
        } else {
            float init = f.init_f;
            if (f.arg >= 0 && (size_t)f.arg < x.n.decl->args.size()) init = x.n.decl->args[f.arg];
            st.push_back(x.state_f(f.name, [init](const UEnv&) { return init; }));
        }
    }
    // EventInstance::frame_offset (graph/types.rs): a handler whose source names `frame_offset` receives the offset of the
    // event inside the process_block call it belongs to, rescaled by N for a node of the oversampled region
    // (compute_event_rescale, ir/lower.rs:846-852; oscen-lib/tests/multirate_graph.rs EventOffsetRescale)
    bool uses_offset = false;
    for (const auto& h : u.handlers) {
        const std::string& hs = h.second;
        for (size_t pos = hs.find("frame_offset"); pos != std::string::npos; pos = hs.find("frame_offset", pos + 1)) {
            const bool lb = pos == 0 || !(isalnum((unsigned char)hs[pos - 1]) || hs[pos - 1] == '_');
            const bool rb = pos + 12 >= hs.size() || !(isalnum((unsigned char)hs[pos + 12]) || hs[pos + 12] == '_');
            uses_offset = uses_offset || (lb && rb);
        }
    }
    if (uses_offset && x.cg.blk_slot0 < 0) {
        for (int k = 0; k < 32; ++k) {
            const int sl = x.cg.new_slot([k](const UEnv& e) { return (uint32_t)k < e.n_blocks && e.block_starts ? e.block_starts[k] : 0xFFFFFFFFu; });
            if (k == 0) x.cg.blk_slot0 = sl;
        }
        x.cg.common_decl << "    auto blk_off = [&](const uint32_t ff) __attribute__((always_inline)) -> uint32_t { // offset inside its block\n"
                         << "        uint32_t s0 = 0u;\n"
                         << "        for (uint32_t k = 1u; k < 32u; ++k) { const uint32_t st = SU(" << x.cg.blk_slot0 << "u + k); s0 = (st <= ff) ? st : s0; }\n"
                         << "        return ff - s0;\n    };\n";
    }
    // the device functions of this type, once per graph
    if (!x.cg.user_fns.count(u.type)) {
        std::ostringstream d;
        auto params = [&](bool handler) {
            std::ostringstream q;
            bool first = true;
            auto sep = [&]() {
                if (!first) q << ", ";
                first = false;
            };
            if (handler) {
                sep();
                q << "const float value";
                if (uses_offset) q << ", const uint32_t frame_offset";
            }
            for (const UserPort& p : u.inputs) {
                if (p.kind == Kind::Event) continue;
                if (handler && p.kind != Kind::Value) continue; // a handler runs before the frame's streams exist
                sep();
                if (p.channels > 1) q << "const og::Frame<" << p.channels << "> " << p.name;
                else q << "const float " << p.name;
            }
            for (const UserState& f : u.state) {
                sep();
                q << (f.is_uint ? "uint32_t& " : "float& ") << f.name;
            }
            if (!handler)
                for (size_t oi = 0; oi < u.outputs.size(); ++oi) {
                    sep();
                    const int w = oi < u.out_channels.size() ? u.out_channels[oi] : 1;
                    if (w > 1) q << "og::Frame<" << w << ">& " << u.outputs[oi];
                    else q << "float& " << u.outputs[oi];
                }
            for (const std::string& o : u.ev_outputs) {
                sep();
                q << "og::EvOut& " << o;
            }
            sep();
            q << "const float sample_rate";
            return q.str();
        };
        d << "// user node type " << u.type << " (og_register_node)\n"
          << "__device__ __forceinline__ void " << fn << "_process(" << params(false) << ")\n{\n"
          << u.process_src << "\n}\n";
        for (const auto& h : u.handlers)
            d << "__device__ __forceinline__ void " << fn << "_on_" << h.first << "(" << params(true) << ")\n{\n" << h.second
              << "\n}\n";
        x.cg.user_fns[u.type] = d.str();
    }
    // event outputs: one register queue each, visible to the handlers (kernel event loop) and to the tick
    std::vector<std::string> evo;
    if (!u.ev_outputs.empty() && u.event_capacity > x.cg.ev_capacity) x.cg.ev_capacity = std::min(32, u.event_capacity);
    for (const std::string& o : u.ev_outputs) {
        evo.push_back(x.p + o);
        x.cg.S().decl << "    og::EvOut " << x.p << o << ";\n";
        const bool to_inner = x.n.domain == 1 && x.cg.inner_ev_sources.count({x.n.id, o}) > 0;
        const bool to_outer = x.n.domain == 1 && x.cg.outer_ev_sources.count({x.n.id, o}) > 0; // collects the frame's N ticks for an outer node
        (to_inner ? x.cg.S().s_ev6a_clear : ((x.n.domain == 1 && !to_outer) ? x.cg.S().s_cap : x.cg.frame_end))
            << "        if (__any((int)(" << x.p << o << ".lost != 0u))) og::ev_report_lost(A, c, " << x.p << o << ");\n"
            << "        " << x.p << o << ".clear();\n";
        if (to_inner) { // the queue outlives the frame -- and the launch: it is part of the voice's state
            const int wn = x.cg.new_state(x.n.decl->name + "." + o + ".n", false, [](const UEnv&) { return 0u; });
            x.cg.S().load << "        " << x.p << o << ".n = og::ld_u(A, c, " << wn << ");\n";
            x.cg.S().store << "        og::st_u(A, c, " << wn << ", " << x.p << o << ".n);\n";
            bool sets_offsets = u.process_src.find("push_at") != std::string::npos;
            for (const auto& h : u.handlers) sets_offsets = sets_offsets || h.second.find("push_at") != std::string::npos;
            for (int k = 0; k < x.cg.ev_capacity; ++k) {
                const int wv = x.cg.new_state(x.n.decl->name + "." + o + ".v" + std::to_string(k), true, [](const UEnv&) { return fbits(0.0f); });
                x.cg.S().load << "        " << x.p << o << ".v[" << k << "] = og::ld_f(A, c, " << wv << ");\n";
                x.cg.S().store << "        og::st_f(A, c, " << wv << ", " << x.p << o << ".v[" << k << "]);\n";
                if (!sets_offsets) continue; // (every push is push(x): all offsets are 0)
                const int wo = x.cg.new_state(x.n.decl->name + "." + o + ".o" + std::to_string(k), false, [](const UEnv&) { return 0u; });
                x.cg.S().load << "        " << x.p << o << ".o[" << k << "] = og::ld_u(A, c, " << wo << ");\n";
                x.cg.S().store << "        og::st_u(A, c, " << wo << ", " << x.p << o << ".o[" << k << "]);\n";
            }
        }
        x.cg.out.has_node_event_outputs = true;
    }
    // event handlers: value inputs must be known when the event fires (before the frame's nodes run)
    for (const auto& h : u.handlers) {
        if (!x.n.ev_edges.count(h.first) && !x.n.ev_node_edges.count(h.first)) continue;
        const bool from_node = x.n.ev_node_edges.count(h.first) > 0;
        std::ostringstream args;
        for (size_t i = 0; i < ins.size(); ++i) {
            if (in_ports[i]->kind != Kind::Value) continue;
            if (!from_node && (ins[i].rate == Rate::Vary || ins[i].rate == Rate::UFrame))
                fail("node '" + x.n.decl->name + "' (" + u.type + "): value input '" + in_ports[i]->name +
                     "' must be constant over the block because the node has an event handler");
            args << ", " << ins[i].e;
        }
        for (const std::string& v : st) args << ", " << v;
        for (const std::string& v : evo) args << ", " << v;
        args << ", " << x.sf(s_sr) << ");\n";
        const std::string tail = args.str(), name = fn + "_on_" + h.first;
        const std::string off = !uses_offset ? std::string() : std::string(", @OFF@"); // (NodeCtx::on_event: by the event's source)
        x.on_event(h.first, [&](const std::string& val) { return "                " + name + "(" + val + off + tail; });
    }
    for (const auto& kv : x.n.ev_edges)
        if (!u.handlers.count(kv.first))
            fail("node '" + x.n.decl->name + "' (" + u.type + "): event input '" + kv.first + "' has no on_" + kv.first + " handler");
    for (const auto& kv : x.n.ev_node_edges)
        if (!u.handlers.count(kv.first))
            fail("node '" + x.n.decl->name + "' (" + u.type + "): event input '" + kv.first + "' has no on_" + kv.first + " handler");
    // the tick
    std::ostringstream call;
    auto out_width = [&](size_t oi) { return oi < u.out_channels.size() ? u.out_channels[oi] : 1; };
    for (size_t oi = 0; oi < u.outputs.size(); ++oi) {
        if (out_width(oi) > 1) x.cg.os() << "        og::Frame<" << out_width(oi) << "> " << x.p << u.outputs[oi] << " = {};\n";
        else x.cg.os() << "        float " << x.p << u.outputs[oi] << " = 0.0f;\n";
    }
    call << "        " << fn << "_process(";
    bool first = true;
    auto sep = [&]() {
        if (!first) call << ", ";
        first = false;
    };
    for (const Val& v : ins) {
        sep();
        call << v.e;
    }
    for (const std::string& v : st) {
        sep();
        call << v;
    }
    for (const std::string& o : u.outputs) {
        sep();
        call << x.p << o;
    }
    for (const std::string& v : evo) {
        sep();
        call << v;
    }
    sep();
    call << x.sf(s_sr) << ");\n";
    x.cg.os() << call.str();
    for (size_t oi = 0; oi < u.outputs.size(); ++oi)
        if (out_width(oi) > 1) { // the channels of a Frame<N> output continue as scalar values "<port>#i"
            std::vector<std::string> chans;
            for (int c = 0; c < out_width(oi); ++c) chans.push_back(x.p + u.outputs[oi] + ".v[" + std::to_string(c) + "]");
            x.set_out_frame(u.outputs[oi], chans);
        }
    for (size_t oi = 0; oi < u.outputs.size(); ++oi) {
        if (out_width(oi) > 1) continue;
        const std::string& o = u.outputs[oi];
        Val v;
        v.e = x.p + o;
        v.rate = Rate::Vary;
        v.inner = x.n.domain == 1;
        const std::string key = "n" + std::to_string(x.n.id) + "." + o;
        x.cg.node_outputs[key] = v;
        auto fit = x.cg.fb_vars.find(key);
        if (fit != x.cg.fb_vars.end()) x.cg.os() << "        " << fit->second << " = " << v.e << ";\n";
    }
}

const NodeTypeInfo* lookup_type(const std::string& type_in)
{
    const std::string type = normalize_type(type_in);
    auto it = registry().find(type);
    if (it != registry().end()) return &it->second;
    auto ut = user_registry().find(type);
    return ut == user_registry().end() ? nullptr : &ut->second->info;
}

int user_weight(const std::string& type)
{
    auto ut = user_registry().find(type);
    return ut == user_registry().end() ? 0 : ut->second->t.weight;
}

// ---- node arrays and nested graphs: desugared before lowering ------------------------------------------
std::map<std::string, GraphDesc>& graph_types()
{
    // starts out with the reference's own voice graphs (og_builtin.cpp), so that the example crates' poly wrappers
    // (`voices = [FMVoice::new(); 8]`) lower without registration; og_register_graph_type may replace them
    static std::map<std::string, GraphDesc> R = builtin_voice_graph_types();
    return R;
}

struct Tok {
    enum K { Ident, Other } k;
    std::string text;  // identifier, or the raw text of everything else
    long index = -1;   // `ident[index]`
    std::string port;  // `.port` following the identifier (and its index)
    bool call = false; // legacy `.port()` accessor
};

// split an endpoint expression into identifier references (with optional [i] and .port) and the text between them
std::vector<Tok> scan(const std::string& s)
{
    std::vector<Tok> out;
    size_t i = 0;
    auto other = [&](char c) {
        if (out.empty() || out.back().k != Tok::Other) out.push_back({Tok::Other, ""});
        out.back().text.push_back(c);
    };
    while (i < s.size()) {
        const char c = s[i];
        if (isalpha((unsigned char)c) || c == '_') {
            Tok t{Tok::Ident, ""};
            const size_t start = i;
            while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_')) t.text.push_back(s[i++]);
            size_t j = i;
            while (j < s.size() && isspace((unsigned char)s[j])) ++j;
            // not a reference: a method name (`.tanh(`), a function name (`half(`) or a segment of its path (`dsp::`)
            size_t b = start;
            while (b > 0 && isspace((unsigned char)s[b - 1])) --b;
            const bool after_dot = b > 0 && s[b - 1] == '.', after_path = b > 1 && s[b - 1] == ':' && s[b - 2] == ':';
            const bool before_call = j < s.size() && s[j] == '(', before_path = j + 1 < s.size() && s[j] == ':' && s[j + 1] == ':';
            if (after_dot || before_call || before_path || after_path) {
                for (char ch : t.text) other(ch);
                continue;
            }
            if (j < s.size() && s[j] == '[') {
                size_t k = j + 1;
                std::string num;
                while (k < s.size() && s[k] != ']') num.push_back(s[k++]);
                if (k >= s.size()) fail("missing ']' in '" + s + "'");
                t.index = strtol(num.c_str(), nullptr, 10);
                i = k + 1;
                j = i;
                while (j < s.size() && isspace((unsigned char)s[j])) ++j;
            }
            if (j < s.size() && s[j] == '.' && j + 1 < s.size() && (isalpha((unsigned char)s[j + 1]) || s[j + 1] == '_')) {
                size_t k = j + 1;
                std::string port;
                while (k < s.size() && (isalnum((unsigned char)s[k]) || s[k] == '_')) port.push_back(s[k++]);
                size_t q = k;
                while (q < s.size() && isspace((unsigned char)s[q])) ++q;
                const bool paren = q < s.size() && s[q] == '(';
                const bool accessor = paren && q + 1 < s.size() && s[q + 1] == ')' && !Parser::is_method(port); // legacy `osc.output()`
                if (!paren || accessor) {
                    t.port = port;
                    i = k;
                    if (accessor) {
                        t.call = true;
                        i = q + 2;
                    }
                } // else `input.abs()`: a method on a bare identifier -- the `.abs(` stays text
            }
            out.push_back(t);
        } else if (isdigit((unsigned char)c) || (c == '.' && i + 1 < s.size() && isdigit((unsigned char)s[i + 1]))) {
            // a numeric literal (may carry a rust suffix such as f32): not an identifier
            while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '.' || s[i] == '_' ||
                                    ((s[i] == '+' || s[i] == '-') && (s[i - 1] == 'e' || s[i - 1] == 'E'))))
                other(s[i++]);
        } else {
            other(c);
            ++i;
        }
    }
    return out;
}

std::string unscan(const std::vector<Tok>& t)
{
    std::string r;
    for (const Tok& k : t) {
        r += k.text;
        if (k.k == Tok::Ident) {
            if (k.index >= 0) r += "[" + std::to_string(k.index) + "]";
            if (!k.port.empty()) r += "." + k.port;
        }
    }
    return r;
}

std::string elem_name(const std::string& arr, long i) { return arr + "__" + std::to_string(i); }

// `name = [Type::ctor(..); N]` -> N nodes; edges by classify_fanout (ir/lower.rs:784-786, codegen/emit_edge.rs:30-84):
// array -> array of the same length = parallel, scalar -> array = broadcast, array -> scalar = sum in index order
GraphDesc expand_arrays(const GraphDesc& g)
{
    std::map<std::string, uint32_t> arr;
    for (const GNode& n : g.nodes)
        if (n.array_len) arr[n.name] = n.array_len;
    // `<name>__<digits>` is how the elements of an expanded array are named, and the endpoint-kind inference further down
    // shares one kind per array ROOT by stripping that suffix (canon()).  A graph written out element by element
    // (`oscs__0`, `oscs__1`, ...: what to_dsl prints for an expanded array) is fine -- the elements share their root's kinds,
    // as in the array form; a node spelled like an element of ANOTHER node of the graph (`osc__1` next to `osc`) would
    // share kinds with an unrelated node: refused
    {
        std::set<std::string> names;
        for (const GNode& n : g.nodes) names.insert(n.name);
        for (const GNode& n : g.nodes) {
            const size_t us = n.name.rfind("__");
            if (n.array_len || us == std::string::npos || us == 0 || us + 2 >= n.name.size() ||
                !std::all_of(n.name.begin() + (long)us + 2, n.name.end(), [](char ch) { return isdigit((unsigned char)ch); }))
                continue;
            if (names.count(n.name.substr(0, us)))
                fail("node '" + n.name + "': names ending in __<digits> are reserved for the elements of node arrays (the graph also has a node '" +
                     n.name.substr(0, us) + "')");
        }
    }
    if (arr.empty()) return g;
    GraphDesc o;
    o.name = g.name;
    o.inputs = g.inputs;
    o.outputs = g.outputs;
    for (const GNode& n : g.nodes) {
        if (!n.array_len) {
            o.nodes.push_back(n);
            continue;
        }
        for (uint32_t i = 0; i < n.array_len; ++i) {
            GNode e = n;
            e.name = elem_name(n.name, i);
            e.array_len = 0;
            o.nodes.push_back(e);
        }
    }
    for (const GEdge& e : g.edges) {
        std::vector<Tok> src = scan(e.src), dst = scan(e.dst);
        // destination: `node.port`, `node[i].port` or a graph output
        Tok* d = nullptr;
        for (Tok& t : dst)
            if (t.k == Tok::Ident) d = d ? d : &t;
        if (!d) fail("bad destination '" + e.dst + "'");
        uint32_t dst_n = 0;
        if (arr.count(d->text)) {
            if (d->index >= 0) {
                if ((uint32_t)d->index >= arr[d->text]) fail("index out of range in '" + e.dst + "'");
                d->text = elem_name(d->text, d->index);
                d->index = -1;
            } else {
                dst_n = arr[d->text];
            }
        }
        // sources: indexed array elements are plain nodes; un-indexed array references make the edge an array edge
        uint32_t src_n = 0;
        size_t n_refs = 0;
        for (Tok& t : src) {
            if (t.k != Tok::Ident) continue;
            ++n_refs;
            auto it = arr.find(t.text);
            if (it == arr.end()) continue;
            if (t.index >= 0) {
                if ((uint32_t)t.index >= it->second) fail("index out of range in '" + e.src + "'");
                t.text = elem_name(t.text, t.index);
                t.index = -1;
            } else {
                if (src_n && src_n != it->second) fail("array sources of different lengths in '" + e.src + "'");
                src_n = it->second;
            }
        }
        auto with_index = [&](const std::vector<Tok>& toks, long i, bool is_dst) {
            std::vector<Tok> r = toks;
            for (Tok& t : r)
                if (t.k == Tok::Ident && t.index < 0 && arr.count(t.text) && (is_dst ? &t - &r[0] == d - &dst[0] : true))
                    t.text = elem_name(t.text, i);
            return unscan(r);
        };
        if (dst_n) {
            if (src_n && src_n != dst_n)
                fail("array connection '" + e.src + " -> " + e.dst + "': lengths differ (" + std::to_string(src_n) + " vs " +
                     std::to_string(dst_n) + ")");
            for (uint32_t i = 0; i < dst_n; ++i) { // parallel (src_n == dst_n) or broadcast (src_n == 0)
                GEdge x = e;
                x.src = with_index(src, i, false);
                x.dst = with_index(dst, i, true);
                o.edges.push_back(x);
            }
        } else if (src_n) {
            // array -> scalar: `dest = src.iter().map(|n| n.field).sum()` -- only a plain `array.field` source
            size_t n_idents = 0;
            bool plain = true;
            for (const Tok& t : src) {
                if (t.k == Tok::Ident) ++n_idents;
                else
                    for (char c : t.text)
                        if (!isspace((unsigned char)c)) plain = false;
            }
            if (n_idents != 1 || !plain)
                fail("array source inside a compound expression needs an array destination ('" + e.src + " -> " + e.dst + "')");
            if (!e.policy.empty()) {
                // across a rate boundary the elements are summed at the SOURCE rate, in index order, and the sum goes
                // through one resampler (`[sinc] emitters.output -> out` with `emitters = [..; 4] * 2`:
                // codegen/emit_edge.rs:208-236 `__sum`, captured per inner tick, emit_frame.rs:428-458)
                GEdge x = e;
                x.src.clear();
                for (uint32_t i = 0; i < src_n; ++i) x.src += (i ? " + " : "") + with_index(src, i, false);
                x.dst = unscan(dst);
                o.edges.push_back(x);
                continue;
            }
            for (uint32_t i = 0; i < src_n; ++i) { // several sources into one input = their sum in edge (= index) order
                GEdge x = e;
                x.src = with_index(src, i, false);
                x.dst = unscan(dst);
                o.edges.push_back(x);
            }
        } else {
            GEdge x = e;
            x.src = unscan(src);
            x.dst = unscan(dst);
            o.edges.push_back(x);
        }
        (void)n_refs;
    }
    return o;
}

// `sub = SubGraph::new()` (or bare `SubGraph`) with SubGraph a registered graph type: its nodes and connections are
// inlined under the prefix `sub_`; its inputs take what the outer graph connects to `sub.<input>` (or their defaults),
// its outputs stand for the inner expressions that feed them.
GraphDesc expand_nested(const GraphDesc& g, int depth)
{
    if (depth > 8) fail("nested graphs deeper than 8 levels (recursive graph type?)");
    auto type_of = [&](const GNode& n) -> const GraphDesc* {
        std::string t = n.type;
        const size_t c = t.find("::");
        if (c != std::string::npos && t.substr(c) == "::new") t = t.substr(0, c);
        auto it = graph_types().find(t);
        return (it == graph_types().end() || lookup_type(n.type)) ? nullptr : &it->second;
    };
    bool any = false;
    for (const GNode& n : g.nodes) any = any || type_of(n);
    if (!any) return g;
    GraphDesc o;
    o.name = g.name;
    o.inputs = g.inputs;
    o.outputs = g.outputs;
    struct Sub {
        GraphDesc g;                                      // expanded inner graph
        std::map<std::string, std::vector<std::string>> in_src; // inner input -> outer source expressions (edge order)
        std::map<std::string, std::string> in_policy;           // inner input -> connection policy of the outer edge(s)
        std::map<std::string, std::string> out_expr;      // inner output -> expression over (prefixed) inner nodes
    };
    std::map<std::string, Sub> subs;
    for (const GNode& n : g.nodes) {
        const GraphDesc* t = type_of(n);
        if (!t) continue;
        subs[n.name].g = expand_nested(expand_arrays(*t), depth + 1);
        if (n.rate_factor != 1) { // `inner = InnerGraph::new() * 2`: every node of the nested graph runs at that rate
            for (GNode& in : subs[n.name].g.nodes) {
                if (in.rate_factor != 1)
                    fail("nested graph '" + n.name + "' is oversampled as a whole and has oversampled nodes of its own ('" + in.name + "')");
                in.rate_factor = n.rate_factor;
            }
        }
    }
    // outer edges into sub-graph inputs
    std::vector<GEdge> outer;
    for (const GEdge& e : g.edges) {
        std::vector<Tok> dst = scan(e.dst);
        const Tok* d = nullptr;
        for (const Tok& t : dst)
            if (t.k == Tok::Ident) d = d ? d : &t;
        if (d && subs.count(d->text) && !d->port.empty()) {
            if (e.feedback) fail("feedback edges cannot target a nested graph input ('" + e.dst + "')");
            subs[d->text].in_src[d->port].push_back(e.src);
            // `[policy] src -> inner.x` (an oversampled nested graph): the policy goes to the inner edges that read x
            if (!e.policy.empty()) {
                std::string& pol = subs[d->text].in_policy[d->port];
                if (!pol.empty() && pol != e.policy) fail("nested graph input '" + e.dst + "' is fed with different connection policies");
                pol = e.policy;
            }
            continue;
        }
        outer.push_back(e);
    }
    // inline every sub-graph
    for (const GNode& n : g.nodes) {
        auto sit = subs.find(n.name);
        if (sit == subs.end()) {
            o.nodes.push_back(n);
            continue;
        }
        Sub& sb = sit->second;
        const std::string pre = n.inline_bare ? std::string() : n.name + "_";
        std::map<std::string, const GInput*> inner_in;
        for (const GInput& in : sb.g.inputs) inner_in[in.name] = &in;
        std::map<std::string, bool> inner_node;
        for (const GNode& in : sb.g.nodes) inner_node[in.name] = true;
        for (const auto& kv : sb.in_src)
            if (!inner_in.count(kv.first)) fail("nested graph '" + n.name + "' has no input '" + kv.first + "'");
        for (GNode in : sb.g.nodes) {
            if (in.bus) fail("a nested graph cannot contain post-mix (bus) nodes");
            in.name = pre + in.name;
            o.nodes.push_back(in);
        }
        // rewrite an inner expression: nodes get the prefix, inputs become the outer sources (or the default)
        std::string carried_policy; // policy of the outer edge into the input the expression being rewritten reads
        std::set<std::string> pass_made; // unit-gain "input field" nodes created for compound reads of policy-fed inputs
        std::vector<GEdge> pass_edges;
        auto rewrite = [&](const std::string& expr, bool& is_event, std::string& event_src) {
            std::vector<Tok> t = scan(expr);
            carried_policy.clear();
            for (Tok& k : t) {
                if (k.k != Tok::Ident) continue;
                if (inner_node.count(k.text)) {
                    k.text = pre + k.text;
                    continue;
                }
                auto iit = inner_in.find(k.text);
                if (iit == inner_in.end()) continue;
                const GInput& gi = *iit->second;
                {
                    auto pit = sb.in_policy.find(gi.name);
                    if (pit != sb.in_policy.end()) {
                        size_t idents = 0;
                        bool plain = true;
                        for (const Tok& q : t) {
                            if (q.k == Tok::Ident) ++idents;
                            else
                                for (char ch : q.text) plain = plain && isspace((unsigned char)ch);
                        }
                        if (idents != 1 || !plain) {
                            // read inside a compound expression (`half(x) * 3.0 -> c.input`): the reference resamples the
                            // outer source into the inner graph's input FIELD and the expression reads the field.  The
                            // field here is a unit-gain node of the nested graph's rate (x * 1.0 is x, bit for bit) fed
                            // through the policy; the expression reads its output.
                            const std::string pass = pre + "in__" + gi.name;
                            if (!pass_made.count(pass)) {
                                auto src0 = sb.in_src.find(gi.name);
                                if (src0 == sb.in_src.end()) fail("internal: policy without a source for '" + gi.name + "'");
                                GNode pn;
                                pn.name = pass;
                                pn.type = "Gain::new";
                                pn.args = {1.0f};
                                pn.rate_factor = n.rate_factor;
                                o.nodes.push_back(pn);
                                GEdge pe;
                                for (size_t i = 0; i < src0->second.size(); ++i) pe.src += (i ? " + (" : "(") + src0->second[i] + ")";
                                pe.dst = pass + ".input";
                                pe.policy = pit->second;
                                pass_edges.push_back(pe);
                                pass_made.insert(pass);
                            }
                            k = Tok{Tok::Other, "(" + pass + ".output)"};
                            continue;
                        }
                        carried_policy = pit->second;
                    }
                }
                auto src = sb.in_src.find(gi.name);
                if (gi.kind == Kind::Event) {
                    is_event = true;
                    if (src == sb.in_src.end()) {
                        event_src.clear();
                    } else {
                        if (src->second.size() != 1) fail("event input '" + n.name + "." + gi.name + "' needs exactly one source");
                        event_src = src->second[0];
                    }
                    continue;
                }
                // A `[ramp: N]` input of a nested graph is a ValueRampState field of the nested struct, ticked by the nested
                // process() itself (codegen/mod.rs:559-572).  Nothing in the outer graph can move it: an outer edge into it
                // has no ConnectEndpoints<f32, ValueRampState> impl (graph/static_context.rs:41-147: a Rust type error), so
                // it stays at its default -- an idle ramp, `current` = the declared value on every tick.
                if (gi.ramp_frames && src != sb.in_src.end())
                    fail("nested graph '" + n.name + "': '" + gi.name + "' is a ramped input (a ValueRampState): an outer connection "
                         "cannot drive it (the reference has no ConnectEndpoints impl for it)");
                std::string rep;
                if (src == sb.in_src.end()) {
                    rep = flit(gi.def);
                    rep = std::to_string(0) == "0" ? rep : rep;
                    char buf[64];
                    snprintf(buf, sizeof buf, "%.9g", (double)gi.def);
                    rep = buf;
                } else {
                    for (size_t i = 0; i < src->second.size(); ++i) rep += (i ? " + (" : "(") + src->second[i] + ")";
                }
                k = Tok{Tok::Other, "(" + rep + ")"};
            }
            return unscan(t);
        };
        for (const GEdge& ie : sb.g.edges) {
            bool is_event = false;
            std::string event_src;
            GEdge x = ie;
            x.src = rewrite(ie.src, is_event, event_src);
            if (!carried_policy.empty()) {
                if (!x.policy.empty() && x.policy != carried_policy)
                    fail("nested graph '" + n.name + "': edge '" + ie.src + " -> " + ie.dst + "' already has a connection policy");
                x.policy = carried_policy;
            }
            if (is_event) {
                if (event_src.empty()) continue; // unconnected event input: the inner handlers never fire
                x.src = event_src;
            }
            std::vector<Tok> dst = scan(ie.dst);
            bool to_output = true;
            for (Tok& k : dst)
                if (k.k == Tok::Ident && inner_node.count(k.text)) {
                    k.text = pre + k.text;
                    to_output = false;
                }
            if (to_output) { // feeds an inner output: remember the expression, no edge
                std::string& oe = sb.out_expr[ie.dst];
                oe = oe.empty() ? "(" + x.src + ")" : oe + " + (" + x.src + ")";
                continue;
            }
            x.dst = unscan(dst);
            o.edges.push_back(x);
        }
        for (const GEdge& pe : pass_edges) outer.push_back(pe); // (their sources are outer expressions: rewritten below)
    }
    // outer edges: `sub.out` stands for the inner expression
    for (GEdge e : outer) {
        std::vector<Tok> t = scan(e.src);
        for (Tok& k : t) {
            if (k.k != Tok::Ident || !subs.count(k.text) || k.port.empty()) continue;
            auto oe = subs[k.text].out_expr.find(k.port);
            if (oe == subs[k.text].out_expr.end()) fail("nested graph '" + k.text + "' has no (connected) output '" + k.port + "'");
            k = Tok{Tok::Other, "(" + oe->second + ")"};
        }
        e.src = unscan(t);
        o.edges.push_back(e);
    }
    return o;
}

} // namespace

// `TptFilter::<Frame<2>>::new` / `::<Stereo>` / `::<Quad>` / `::<f32>` / `::<Mono>` -> "TptFilter<2>::new" (f32: "TptFilter::new")
std::string normalize_type(const std::string& type)
{
    const size_t a = type.find("::<");
    if (a == std::string::npos) return type;
    const size_t b = type.rfind(">::");
    if (b == std::string::npos || b < a) return type;
    std::string g = type.substr(a + 3, b - (a + 3));
    g.erase(std::remove_if(g.begin(), g.end(), [](char ch) { return isspace((unsigned char)ch); }), g.end());
    int w = 0;
    if (g == "f32") w = 1;
    else if (g == "Mono") w = 1;
    else if (g == "Stereo") w = 2;
    else if (g == "Quad") w = 4;
    else if (g.rfind("Frame<", 0) == 0 && g.size() >= 8 && g.back() == '>') w = atoi(g.c_str() + 6);
    if (w <= 0) return type;
    return type.substr(0, a) + (w == 1 ? std::string() : "<" + std::to_string(w) + ">") + type.substr(b + 1);
}


uint32_t RingSpec::capacity(float graph_sr) const
{
    const float want = 2.0f * (graph_sr * rate_factor); // delay/mod.rs:62-66: (2.0 * sr) as usize, capped at 88200
    size_t n = want > 0.0f ? (size_t)want : 0;
    n = std::min<size_t>(n, 88200);
    n = std::max<size_t>(n, 1);
    uint32_t c = 1;
    while (c < n) c <<= 1; // RingBuffer::new -> PowerOfTwo mode (ring_buffer/mod.rs:29-41)
    return c;
}


// `pass = EventPassthrough::new()` (oscen-lib/src/event_passthrough.rs: on_input forwards the event to `output`) is
// pure routing: `x -> pass.input; pass.output -> d` delivers x's events to d on the same frame.  The node is removed
// and its consumers are connected to what feeds it (the last source connected to `input`: clear + copy).
GraphDesc expand_passthrough(const GraphDesc& g)
{
    auto strip = [](std::string t) {
        t.erase(std::remove_if(t.begin(), t.end(), [](char ch) { return isspace((unsigned char)ch); }), t.end());
        return t;
    };
    GraphDesc r = g;
    for (int guard = 0; guard < 64; ++guard) {
        size_t pi = r.nodes.size();
        for (size_t i = 0; i < r.nodes.size(); ++i)
            if (r.nodes[i].type == "EventPassthrough::new" || r.nodes[i].type == "EventPassthrough") pi = i;
        if (pi == r.nodes.size()) return r;
        const std::string nm = r.nodes[pi].name;
        std::string feed; // the source that ends up in pass.input
        for (const GEdge& e : r.edges)
            if (strip(e.dst) == nm + ".input") feed = e.src;
        if (strip(feed) == nm + ".output") fail("event passthrough '" + nm + "' feeds itself");
        std::vector<GEdge> edges;
        for (const GEdge& e : r.edges) {
            if (strip(e.dst) == nm + ".input") continue;
            if (strip(e.src) == nm + ".output") {
                if (feed.empty()) continue; // nothing ever arrives
                GEdge f = e;
                f.src = feed;
                edges.push_back(f);
            } else {
                if (strip(e.src).find(nm + ".") == 0 || strip(e.dst).find(nm + ".") == 0)
                    fail("EventPassthrough '" + nm + "' has the ports 'input' and 'output' only ('" + e.src + " -> " + e.dst + "')");
                edges.push_back(e);
            }
        }
        r.edges = edges;
        r.nodes.erase(r.nodes.begin() + (long)pi);
    }
    fail("event passthrough chain too long");
}

// ---- poly wrapper graphs -----------------------------------------------------------------------------------
// `midi_parser = MidiParser::new(); voice_allocator = VoiceAllocator::<N>::new(); voice_handlers =
// [MidiVoiceHandler::new(); N]; voices = [Voice::new(); N]; [tremolo = Tremolo::new();]` with the wiring of
// examples/fm-synth/src/lib.rs:68-131 / examples/electric-piano/src/main.rs:52-96.  On this engine the three MIDI
// node kinds are the host-side front end (og_midi_*: same parser / allocator / handler decisions, N = the bank size),
// the voice array IS the bank, `voices.out -> out` is the mix bus and a node fed by the voice sum is the post-mix
// stage.  The description is rewritten into the voice-bank graph the engine runs:
//   inputs   <handler.frequency target> (per voice), <handler.gate target> (event), then the wrapper's value inputs
//            (defaults and [ramp: N] kept); the raw-MIDI event input is dropped (og_midi_send takes its place)
//   nodes    the voice graph's own nodes under their own names (prefixed with `<array>_` only if a name collides),
//            wrapper parameters wired to the voice inputs exactly as `param -> voices.param` says
//   outputs  the wrapper's stream outputs; with a post-mix node also the voice graph's output (the summed voices)
namespace {
std::string strip_ws(std::string t)
{
    t.erase(std::remove_if(t.begin(), t.end(), [](char ch) { return isspace((unsigned char)ch); }), t.end());
    return t;
}
bool type_is(const GNode& n, const char* base)
{
    const std::string t = n.type;
    const size_t len = strlen(base);
    return t.compare(0, len, base) == 0 && (t.size() == len || t[len] == ':' || t[len] == '<');
}
} // namespace

GraphDesc lower_poly_wrapper(const GraphDesc& g, PolyInfo* info)
{
    const GNode *parser = nullptr, *alloc = nullptr, *handlers = nullptr;
    for (const GNode& n : g.nodes) {
        if (type_is(n, "MidiParser")) parser = &n;
        else if (type_is(n, "VoiceAllocator")) alloc = &n;
        else if (type_is(n, "MidiVoiceHandler")) handlers = &n;
    }
    if (!parser && !alloc && !handlers) return g;
    if (!handlers) fail("poly wrapper: MidiParser / VoiceAllocator without a [MidiVoiceHandler::new(); N] array");
    std::set<std::string> control;
    for (const GNode* n : {parser, alloc, handlers})
        if (n) control.insert(n->name);
    auto root = [](const std::string& s) {
        const std::string t = strip_ws(s);
        return t.substr(0, t.find_first_of(".[("));
    };
    auto port = [](const std::string& s) {
        const std::string t = strip_ws(s);
        const size_t d = t.find('.');
        if (d == std::string::npos) return std::string();
        std::string p = t.substr(d + 1);
        const size_t par = p.find('(');
        return par == std::string::npos ? p : p.substr(0, par);
    };
    // the voice array: what the handlers' frequency / gate outputs feed
    std::string voices, freq_in, gate_in;
    for (const GEdge& e : g.edges) {
        if (root(e.src) != handlers->name) continue;
        const std::string sp = port(e.src), dn = root(e.dst), dp = port(e.dst);
        if (control.count(dn)) continue;
        if (sp == "frequency") {
            voices = voices.empty() ? dn : voices;
            freq_in = dp;
        } else if (sp == "gate") {
            voices = voices.empty() ? dn : voices;
            gate_in = dp;
        } else {
            fail("poly wrapper: MidiVoiceHandler has the outputs 'frequency' and 'gate' ('" + e.src + "')");
        }
        if (dn != voices) fail("poly wrapper: the voice handlers feed more than one node array ('" + dn + "' and '" + voices + "')");
    }
    if (voices.empty() || gate_in.empty())
        fail("poly wrapper: `" + handlers->name + ".gate -> <voices>.<gate input>` is missing");
    const GNode* varr = nullptr;
    for (const GNode& n : g.nodes)
        if (n.name == voices) varr = &n;
    if (!varr) fail("poly wrapper: unknown node '" + voices + "'");
    std::string vtype = varr->type;
    if (vtype.size() > 5 && vtype.compare(vtype.size() - 5, 5, "::new") == 0) vtype.erase(vtype.size() - 5);
    auto git = graph_types().find(vtype);
    if (git == graph_types().end())
        fail("poly wrapper: the voice type '" + vtype + "' is not a graph type (register the voice graph with og_register_graph_type, "
             "or use the built-in FMVoice / ElectricPianoVoiceNode)");
    const GraphDesc& vt = git->second;
    if (handlers->array_len && varr->array_len && handlers->array_len != varr->array_len)
        fail("poly wrapper: " + std::to_string(handlers->array_len) + " voice handlers for " + std::to_string(varr->array_len) + " voices");
    const GInput *vfreq = nullptr, *vgate = nullptr;
    for (const GInput& in : vt.inputs) {
        if (in.name == freq_in) vfreq = &in;
        if (in.name == gate_in) vgate = &in;
    }
    if (!vgate || vgate->kind != Kind::Event) fail("poly wrapper: '" + vtype + "." + gate_in + "' is not an event input");
    if (!freq_in.empty() && (!vfreq || vfreq->kind != Kind::Value)) fail("poly wrapper: '" + vtype + "." + freq_in + "' is not a value input");

    GraphDesc o;
    o.name = g.name;
    PolyInfo pi;
    pi.is_wrapper = true;
    pi.declared_voices = varr->array_len ? varr->array_len : 1;
    pi.voice_type = vtype;
    pi.frequency_input = freq_in;
    pi.gate_input = gate_in;
    // inputs
    std::set<std::string> taken;
    if (vfreq) {
        GInput f = *vfreq;
        f.per_voice = true;
        f.ramp_frames = 0;
        o.inputs.push_back(f);
        taken.insert(f.name);
    }
    {
        GInput ev = *vgate;
        o.inputs.push_back(ev);
        taken.insert(ev.name);
    }
    for (const GInput& in : g.inputs) {
        if (in.kind == Kind::Event) { // the raw MIDI input: every use must be the parser's
            for (const GEdge& e : g.edges)
                if (root(e.src) == in.name && !(parser && root(e.dst) == parser->name))
                    fail("poly wrapper: event input '" + in.name + "' may only feed the MidiParser");
            pi.midi_input = in.name;
            continue;
        }
        if (taken.count(in.name)) fail("poly wrapper: input '" + in.name + "' collides with the per-voice input of the same name");
        o.inputs.push_back(in);
        taken.insert(in.name);
    }
    // the voice graph as ONE nested node, inlined under its own node names unless one collides with a wrapper name
    bool bare = true;
    for (const GNode& vn : vt.nodes) {
        if (taken.count(vn.name)) bare = false;
        for (const GNode& wn : g.nodes)
            if (!control.count(wn.name) && wn.name != voices && wn.name == vn.name) bare = false;
        for (const GOutput& wo : g.outputs)
            if (wo.name == vn.name) bare = false;
    }
    GNode vnode;
    vnode.name = voices;
    vnode.type = vtype + "::new";
    vnode.rate_factor = varr->rate_factor;
    vnode.inline_bare = bare;
    // a node fed by the voice sum = the post-mix stage
    std::set<std::string> post;
    for (const GEdge& e : g.edges)
        if (root(e.src) == voices && !port(e.dst).empty() && !control.count(root(e.dst)) && root(e.dst) != voices) post.insert(root(e.dst));
    // outputs: the wrapper's stream outputs; event outputs fed by the MIDI nodes stay on the host
    std::string sum_output; // graph output that carries the summed voices into the post-mix node
    std::set<std::string> dropped;
    for (const GOutput& out : g.outputs) {
        if (out.kind == Kind::Event) {
            dropped.insert(out.name);
            pi.dropped_event_outputs.push_back(out.name);
        }
    }
    for (const GEdge& e : g.edges)
        if (dropped.count(strip_ws(e.dst)) && !control.count(root(e.src)))
            fail("poly wrapper: event output '" + e.dst + "' is fed by '" + e.src + "' (only the MIDI nodes' events are kept on the host)");
    if (!post.empty()) {
        std::string vout;
        for (const GEdge& e : g.edges)
            if (root(e.src) == voices && post.count(root(e.dst))) vout = port(e.src);
        sum_output = vout;
        for (const GOutput& out : g.outputs)
            if (out.name == sum_output) sum_output = "__voice_sum";
        for (const GNode& vn : vt.nodes)
            if (bare && vn.name == sum_output) sum_output = "__voice_sum";
        GOutput so;
        so.name = sum_output;
        so.kind = Kind::Stream;
        o.outputs.push_back(so);
    }
    for (const GOutput& out : g.outputs)
        if (!dropped.count(out.name)) o.outputs.push_back(out);
    o.nodes.push_back(vnode);
    for (const GNode& n : g.nodes) {
        if (control.count(n.name) || n.name == voices) continue;
        GNode c = n;
        if (post.count(n.name)) c.bus = true;
        o.nodes.push_back(c);
    }
    // edges
    if (vfreq) o.edges.push_back({freq_in, voices + "." + freq_in, ""});
    o.edges.push_back({gate_in, voices + "." + gate_in, ""});
    bool summed = false;
    for (const GEdge& e : g.edges) {
        const std::string sr = root(e.src), dr = root(e.dst);
        if (control.count(sr) || control.count(dr)) continue;      // MIDI plumbing: host side
        if (!pi.midi_input.empty() && sr == pi.midi_input) continue;
        if (dropped.count(strip_ws(e.dst))) continue;
        GEdge x = e;
        if (sr == voices && post.count(dr)) { // voices.output -> tremolo.input  ==>  voices.output -> <sum>; <sum> -> tremolo.input
            if (!summed) o.edges.push_back({strip_ws(e.src), sum_output, e.policy});
            summed = true;
            x.src = sum_output;
            x.policy.clear();
        }
        o.edges.push_back(x);
    }
    if (info) *info = pi;
    return o;
}

GraphDesc expand(const GraphDesc& g) { return expand_passthrough(expand_nested(expand_arrays(lower_poly_wrapper(g)), 0)); }

void register_user_function(const UserFunction& f)
{
    if (f.name.empty()) fail("function without a name");
    size_t b = 0;
    for (;;) { // every path segment an identifier
        const size_t c = f.name.find("::", b);
        if (!is_ident(f.name.substr(b, c == std::string::npos ? std::string::npos : c - b))) fail("function name '" + f.name + "' is not a path of identifiers");
        if (c == std::string::npos) break;
        b = c + 2;
    }
    const std::string last = f.name.substr(b);
    if (last == "Frame") fail("'Frame' is the frame constructor");
    if (f.arg_names.size() != f.arg_channels.size() || f.arg_names.empty() || f.arg_names.size() > 8)
        fail("function '" + f.name + "': 1 to 8 arguments, one width each");
    for (size_t k = 0; k < f.arg_names.size(); ++k) {
        if (!is_ident(f.arg_names[k])) fail("function '" + f.name + "': bad argument name '" + f.arg_names[k] + "'");
        if (f.arg_channels[k] < 1 || f.arg_channels[k] > 4) fail("function '" + f.name + "': an argument is an f32 (1) or a Frame<2..4>");
    }
    if (f.result_channels < 1 || f.result_channels > 4) fail("function '" + f.name + "': the result is an f32 (1) or a Frame<2..4>");
    if (f.source.empty()) fail("function '" + f.name + "' has no body");
    function_registry()[f.name] = f;
}
bool unregister_user_function(const std::string& name) { return function_registry().erase(name) > 0; }

void register_user_node(const UserNodeType& t)
{
    if (t.type.empty()) fail("node type needs a name");
    if (registry().count(t.type)) fail("'" + t.type + "' is a built-in node type");
    // (no outputs at all = a sink, `StereoSink { #[input(stream)] input, last }` of the reference's fixtures: its effect
    // is the state it keeps; ir/passes/dead_nodes.rs keeps such nodes when the graph declares no outputs)
    std::set<std::string> names;
    auto uniq = [&](const std::string& n) {
        if (!is_ident(n)) fail("node type '" + t.type + "': '" + n + "' is not an identifier");
        if (n == "value" || n == "sample_rate" || n == "frame_offset") // (parameters the generated handlers / process() receive)
            fail("node type '" + t.type + "': '" + n + "' is reserved");
        if (!names.insert(n).second) fail("node type '" + t.type + "': duplicate field '" + n + "'");
    };
    for (const auto& p : t.inputs) {
        uniq(p.name);
        if (p.arg >= (int)t.nargs) fail("node type '" + t.type + "': input '" + p.name + "' refers to a missing constructor argument");
    }
    for (const auto& o : t.outputs) uniq(o);
    for (const auto& o : t.ev_outputs) uniq(o);
    for (const auto& f : t.state) {
        uniq(f.name);
        if (f.arg >= (int)t.nargs) fail("node type '" + t.type + "': field '" + f.name + "' refers to a missing constructor argument");
    }
    for (const auto& h : t.handlers) {
        bool ok = false;
        for (const auto& p : t.inputs) ok = ok || (p.kind == Kind::Event && p.name == h.first);
        if (!ok) fail("node type '" + t.type + "': handler for '" + h.first + "', which is not an event input");
    }
    std::unique_ptr<UserEntry> e(new UserEntry);
    e->t = t;
    if (e->t.weight <= 0) { // rough VALU estimate from the source: arithmetic operators, calls weigh more
        int w = 0;
        const std::string& src = e->t.process_src;
        for (size_t i = 0; i < src.size(); ++i) {
            const char c = src[i];
            if (c == '*' || c == '+' || c == '-' || c == '/' || c == '?' || c == '<' || c == '>') ++w;
            if (c == '(' && i > 0 && (isalnum((unsigned char)src[i - 1]) || src[i - 1] == '_')) w += 4;
        }
        e->t.weight = std::max(1, w);
    }
    for (const auto& p : e->t.inputs) {
        if (p.channels > 1 && p.kind != Kind::Stream) fail("node type '" + t.type + "': only stream inputs can be Frame<N> ('" + p.name + "')");
        if (p.channels > 4) fail("node type '" + t.type + "': Frame<N> ports support N <= 4 ('" + p.name + "')");
        e->info.inputs.push_back({p.name.c_str(), p.kind, p.def, p.arg, std::max(1, p.channels)});
    }
    for (const auto& o : e->t.outputs) e->info.outputs.push_back(o.c_str());
    for (size_t oi = 0; oi < e->t.outputs.size(); ++oi) {
        const int w = oi < e->t.out_channels.size() ? std::max(1, e->t.out_channels[oi]) : 1;
        if (w > 4) fail("node type '" + t.type + "': Frame<N> ports support N <= 4 ('" + e->t.outputs[oi] + "')");
        e->info.out_channels.push_back(w);
    }
    e->info.ev_outputs = e->t.ev_outputs;
    e->info.emit = emit_user;
    e->info.variant = 0;
    e->info.nargs = t.nargs;
    e->info.user = &e->t;
    user_registry()[t.type] = std::move(e);
}

bool unregister_user_node(const std::string& type) { return user_registry().erase(type) > 0; }

void register_graph_type(const std::string& name, const GraphDesc& g)
{
    if (!is_ident(name)) fail("graph type name '" + name + "' is not an identifier");
    check_reference_rules(g);
    graph_types()[name] = g;
}
bool unregister_graph_type(const std::string& name)
{
    auto it = graph_types().find(name);
    if (it == graph_types().end()) return false;
    const auto builtin = builtin_voice_graph_types(); // a replaced built-in voice type comes back
    auto b = builtin.find(name);
    if (b != builtin.end()) it->second = b->second;
    else graph_types().erase(it);
    return true;
}

// ---- what the reference REFUSES, on the description as written (before arrays, nested graphs and poly wrappers are
// expanded: the reference's macro sees one `graph!` body at a time) ---------------------------------------------------
//  (1) kind compatibility of a connection statement (ir/lower.rs:459-490, types_compatible :1157-1165): endpoint kinds are
//      known for typed graph inputs / outputs and for the two ends of a stream-only policy, and spread along connection
//      statements to a fixpoint (infer_endpoint_types, ir/lower.rs:233-338); where BOTH ends of a statement are known
//      the pair must be Stream->Stream, Value->Value, Event->Event or Value->Stream.
//  (2) auto-summed fan-in (codegen/emit_node.rs:35-125 classify_stream_fanin): >= 2 edges into one destination slot that is
//      not known to be a value / event are summed, and the sum is only defined for same-rate, simple (plain endpoint),
//      scalar (neither end a node array) sources; anything else is a scoped compile error in the reference.
// Diagnostics carry the reference's wording, so that a user sees the message rustc would have shown.
namespace {
enum class RK { Unknown = 0, Stream, Value, Event };
const char* rk_name(RK k) { return k == RK::Stream ? "Stream" : (k == RK::Value ? "Value" : (k == RK::Event ? "Event" : "Unknown")); }
struct RefEndpoint {
    std::string node, endpoint;
    long index = -1;
    bool ok = false;
};
// `ident`, `ident.field`, `ident[k].field`, `ident.field[k]`, `ident[k]`, each optionally followed by `()`
RefEndpoint ref_endpoint(const std::string& text)
{
    RefEndpoint r;
    size_t i = 0;
    auto ws = [&] {
        while (i < text.size() && isspace((unsigned char)text[i])) ++i;
    };
    auto ident = [&] {
        ws();
        const size_t b = i;
        if (i < text.size() && (isalpha((unsigned char)text[i]) || text[i] == '_'))
            while (i < text.size() && (isalnum((unsigned char)text[i]) || text[i] == '_')) ++i;
        return text.substr(b, i - b);
    };
    auto index = [&]() -> bool { // optional `[k]`
        ws();
        if (i >= text.size() || text[i] != '[') return true;
        ++i;
        ws();
        std::string num;
        while (i < text.size() && isdigit((unsigned char)text[i])) num.push_back(text[i++]);
        ws();
        if (num.empty() || i >= text.size() || text[i] != ']' || r.index >= 0) return false;
        ++i;
        r.index = strtol(num.c_str(), nullptr, 10);
        return true;
    };
    r.node = ident();
    if (r.node.empty() || !index()) return r;
    ws();
    if (i < text.size() && text[i] == '.') {
        ++i;
        r.endpoint = ident();
        if (r.endpoint.empty() || !index()) return r;
        ws();
        if (i + 1 < text.size() && text[i] == '(' ) {
            ++i;
            ws();
            if (i >= text.size() || text[i] != ')') return r;
            ++i;
        }
    } else {
        r.endpoint = r.node; // bare: the implicit endpoint of a graph input / output carries the declaration's name
    }
    ws();
    r.ok = i == text.size();
    return r;
}
// `voices[3].output` -> `voices.output` so that the expression parser (which runs after array expansion elsewhere) reads it
std::string drop_node_indices(const std::string& t)
{
    std::string o;
    for (size_t i = 0; i < t.size();) {
        if (t[i] == '[' && !o.empty() && (isalnum((unsigned char)o.back()) || o.back() == '_')) {
            size_t j = i + 1;
            while (j < t.size() && (isdigit((unsigned char)t[j]) || isspace((unsigned char)t[j]))) ++j;
            if (j < t.size() && t[j] == ']') {
                size_t k = j + 1;
                while (k < t.size() && isspace((unsigned char)t[k])) ++k;
                if (k < t.size() && t[k] == '.' && k + 1 < t.size() && (isalpha((unsigned char)t[k + 1]) || t[k + 1] == '_')) {
                    i = j + 1;
                    continue;
                }
            }
        }
        o.push_back(t[i++]);
    }
    return o;
}
} // namespace

void check_reference_rules(const GraphDesc& g)
{
    struct Named {
        int what; // 0 graph input, 1 graph output, 2 node
        uint32_t array_len = 0, rate = 1;
    };
    std::map<std::string, Named> names;
    std::map<std::pair<std::string, std::string>, RK> kinds;
    auto decl = [](Kind k) { return k == Kind::Stream ? RK::Stream : (k == Kind::Value ? RK::Value : RK::Event); };
    for (const GInput& in : g.inputs) {
        names[in.name] = {0, 0, 1};
        kinds[{in.name, in.name}] = decl(in.kind);
    }
    for (const GOutput& o : g.outputs) {
        names[o.name] = {1, 0, 1};
        kinds[{o.name, o.name}] = decl(o.kind);
    }
    for (const GNode& n : g.nodes) {
        if (n.bus) return; // (a lowered wrapper with its post-mix stage: checked when it was written as a wrapper)
        names[n.name] = {2, n.array_len, n.rate_factor ? n.rate_factor : 1u};
        if (n.name.rfind("__inline_delay_", 0) == 0) { // `-> [N] ->`: synth_delay_endpoints (ir/lower.rs:186-206)
            kinds[{n.name, "input"}] = kinds[{n.name, "output"}] = RK::Stream;
            kinds[{n.name, "delay_samples"}] = kinds[{n.name, "feedback"}] = RK::Value;
        }
    }
    auto lookup = [&](const std::string& node, const std::string& ep) {
        auto it = kinds.find({node, ep});
        return it == kinds.end() ? RK::Unknown : it->second;
    };
    // endpoint_kind_of (ir/lower.rs:937-959)
    std::function<RK(const ExprP&)> kind_of = [&](const ExprP& e) -> RK {
        if (!e) return RK::Unknown;
        switch (e->t) {
        case Expr::Num: return RK::Value;
        case Expr::Ref: return names.count(e->node) ? lookup(e->node, e->port.empty() ? e->node : e->port) : RK::Unknown;
        case Expr::Chan: return e->a && e->a->t == Expr::Ref ? kind_of(e->a) : RK::Unknown;
        case Expr::Bin: {
            const RK l = kind_of(e->a), r = kind_of(e->b);
            if (l == RK::Unknown || r == RK::Unknown || l == RK::Event || r == RK::Event) return RK::Unknown;
            return (l == RK::Stream || r == RK::Stream) ? RK::Stream : RK::Value;
        }
        default: return RK::Unknown; // calls, methods, negation: no inference
        }
    };
    struct Stmt {
        std::string src_text, dst_text, policy;
        ExprP src;
        RefEndpoint src_ep, dst_ep; // src_ep.ok: the source is a plain endpoint
    };
    std::vector<Stmt> stmts;
    for (size_t i = 0; i < g.edges.size(); ++i) {
        const GEdge& e = g.edges[i];
        if (e.feedback) continue; // (second leg of `src -> [via] -> dst`: folded into its statement below)
        Stmt st;
        st.src_text = e.src;
        st.dst_text = e.dst;
        st.policy = e.policy;
        if (i + 1 < g.edges.size() && g.edges[i + 1].feedback) st.dst_text = g.edges[i + 1].dst; // the statement's ends (ir/lower.rs:459-490)
        try {
            st.src = Parser(drop_node_indices(e.src)).parse();
        } catch (const std::exception&) {
            continue; // (malformed expressions are reported where the graph is lowered)
        }
        st.src_ep = ref_endpoint(st.src_text);
        st.dst_ep = ref_endpoint(st.dst_text);
        if (!st.dst_ep.ok || !names.count(st.dst_ep.node)) continue;
        if (st.src_ep.ok && !names.count(st.src_ep.node)) st.src_ep.ok = false;
        stmts.push_back(st);
    }
    for (const Stmt& st : stmts) // stream-only policies type both of their ends (only vacant entries)
        if (st.policy == "linear" || st.policy == "sinc" || st.policy == "sinc_iir") {
            kinds.emplace(std::make_pair(st.dst_ep.node, st.dst_ep.endpoint), RK::Stream);
            if (st.src_ep.ok) kinds.emplace(std::make_pair(st.src_ep.node, st.src_ep.endpoint), RK::Stream);
        }
    for (size_t round = 0; round <= stmts.size(); ++round) {
        bool changed = false;
        for (const Stmt& st : stmts) {
            const RK sk = kind_of(st.src);
            if (sk != RK::Unknown && kinds.emplace(std::make_pair(st.dst_ep.node, st.dst_ep.endpoint), sk).second) changed = true;
            const RK dk = lookup(st.dst_ep.node, st.dst_ep.endpoint);
            if (dk != RK::Unknown && st.src_ep.ok && kinds.emplace(std::make_pair(st.src_ep.node, st.src_ep.endpoint), dk).second) changed = true;
        }
        if (!changed) break;
    }
    // (1) kind compatibility, statement by statement
    for (const Stmt& st : stmts) {
        const RK sk = kind_of(st.src), dk = lookup(st.dst_ep.node, st.dst_ep.endpoint);
        if (sk == RK::Unknown || dk == RK::Unknown) continue;
        const bool ok = sk == dk || (sk == RK::Value && dk == RK::Stream);
        if (!ok)
            fail(std::string("Type mismatch in connection: source is ") + rk_name(sk) + " but destination expects " + rk_name(dk) + " ('" +
                 st.src_text + " -> " + st.dst_text + "')");
    }
    // (2) fan-in: the edges (both legs of a `[via]` statement are edges of their own) by destination slot, in edge order
    struct Edge {
        const GEdge* e;
        RefEndpoint dst;
    };
    std::map<std::string, std::vector<Edge>> buckets;
    std::vector<std::string> bucket_order;
    for (const GEdge& e : g.edges) {
        RefEndpoint d = ref_endpoint(e.dst);
        if (!d.ok || !names.count(d.node)) continue;
        const std::string key = d.node + "." + d.endpoint + "#" + std::to_string(d.index);
        if (!buckets.count(key)) bucket_order.push_back(key);
        buckets[key].push_back({&e, d});
    }
    for (const std::string& key : bucket_order) {
        const std::vector<Edge>& b = buckets[key];
        if (b.size() < 2) continue;
        const RefEndpoint& d = b[0].dst;
        const RK dk = lookup(d.node, d.endpoint);
        if (dk == RK::Value || dk == RK::Event) continue; // per-edge path: last write wins / event queues
        struct Src {
            ExprP x;
            RefEndpoint ep;
            std::string primary;
        };
        std::vector<Src> srcs;
        bool any_event = false, parsed = true;
        for (const Edge& ed : b) {
            Src s;
            try {
                s.x = Parser(drop_node_indices(ed.e->src)).parse();
            } catch (const std::exception&) {
                parsed = false;
                break;
            }
            s.ep = ref_endpoint(ed.e->src);
            std::vector<const Expr*> refs;
            collect_refs(s.x, refs);
            for (const Expr* r : refs)
                if (names.count(r->node)) {
                    s.primary = r->node;
                    break;
                }
            if (kind_of(s.x) == RK::Event) any_event = true;
            srcs.push_back(s);
        }
        if (!parsed || any_event) continue;
        const std::string dest_desc = d.endpoint == d.node ? d.node : d.node + "." + d.endpoint;
        for (size_t k = 0; k < b.size(); ++k) {
            const Src& s = srcs[k];
            const char* what = nullptr;
            const Named* sn = s.primary.empty() ? nullptr : &names[s.primary];
            const Named& dn = names[d.node];
            if (sn && sn->rate != dn.rate) what = "a cross-rate edge";
            else if (!(s.ep.ok && names.count(s.ep.node)) || !sn) what = "a compound (non-endpoint) source";
            else if (sn->array_len && dn.array_len) what = "an array (parallel) source";
            else if (!sn->array_len && dn.array_len) what = "a broadcast source";
            else if (sn->array_len && !dn.array_len) what = "an array fan-in source";
            if (what)
                fail(std::string("fan-in summing supports only same-rate scalar/frame stream sources; saw ") + what + " into `" + dest_desc +
                     "` ('" + b[k].e->src + " -> " + b[k].e->dst + "')");
        }
    }
}

std::unique_ptr<CompiledGraph> compile(const GraphDesc& g_in)
{
    check_reference_rules(g_in);
    const GraphDesc g = expand(g_in);
    auto cgp = std::make_unique<CompiledGraph>();
    CompiledGraph& out = *cgp;
    out.name = g.name;
    Codegen cg(g, out);

    // ---- inputs / outputs ----------------------------------------------------
    for (size_t i = 0; i < g.inputs.size(); ++i) {
        const GInput& in = g.inputs[i];
        if (cg.input_by_name.count(in.name)) fail("duplicate input '" + in.name + "'");
        cg.input_by_name[in.name] = (int)i;
        InputInfo info;
        info.decl = in;
        if (in.kind == Kind::Event) {
            info.event_index = out.n_event_inputs++;
        } else if (in.kind == Kind::Value) {
            if (in.per_voice) {
                if (in.ramp_frames) fail("per-voice input '" + in.name + "' cannot be ramped");
                const float d = in.def;
                info.state_word = cg.new_state("in." + in.name, true, [d](const UEnv&) { return fbits(d); });
                cg.common_decl << "    float vin_" << i << " = 0.0f;\n";
                cg.common_load << "        vin_" << i << " = og::ld_f(A, c, " << info.state_word << ");\n";
            } else {
                const int idx = (int)i;
                info.slot = cg.new_slot([idx](const UEnv& e) { return fbits(e.input_values[idx]); });
                if (in.ramp_frames) info.ramp_row = out.n_ramps++;
            }
        } else { // stream input: one per-frame row of the block table, broadcast to every voice
            info.stream_row = -2 - out.n_streams; // final row = n_ramps + k, fixed below once n_ramps is known
            out.n_streams += std::max(1, in.channels);
            out.n_stream_inputs += 1;
        }
        out.inputs.push_back(info);
    }
    for (auto& info : out.inputs)
        if (info.stream_row <= -2) info.stream_row = out.n_ramps + (-2 - info.stream_row);
    for (size_t i = 0; i < g.outputs.size(); ++i) cg.output_by_name[g.outputs[i].name] = (int)i;

    // ---- nodes ------------------------------------------------------------------
    cg.nodes.resize(g.nodes.size());
    for (size_t i = 0; i < g.nodes.size(); ++i) {
        const GNode& nd = g.nodes[i];
        if (cg.node_by_name.count(nd.name) || cg.input_by_name.count(nd.name))
            fail("duplicate name '" + nd.name + "'");
        const NodeTypeInfo* nti = lookup_type(nd.type);
        if (!nti) fail("unknown node type '" + nd.type + "' (node '" + nd.name + "'); custom nodes are added with og_register_node");
        for (size_t a = 0; a < nd.raw_args.size(); ++a)
            if (!nd.raw_args[a].empty())
                fail("node '" + nd.name + "': constructor argument '" + nd.raw_args[a] + "' of " + nd.type + " is not a number");
        if (nd.args.size() != nti->nargs)
            fail("node '" + nd.name + "': " + nd.type + " takes " + std::to_string(nti->nargs) + " arguments");
        if (nd.rate_factor != 1) { // `* N`, N in {2,4,8} (parse.rs:460-488)
            if (nd.rate_factor != 2 && nd.rate_factor != 4 && nd.rate_factor != 8)
                fail("node '" + nd.name + "': rate factor must be 1, 2, 4, or 8");
            if (cg.N != 1 && cg.N != (int)nd.rate_factor)
                fail_unsupported("all oversampled nodes of a graph must share one factor in this version (node '" + nd.name +
                                 "'); the reference: v1 does not support connections between two differently-rated non-default-rate nodes");
            cg.N = (int)nd.rate_factor;
        }
        cg.node_by_name[nd.name] = (int)i;
        cg.nodes[i].decl = &nd;
        cg.nodes[i].type = nti;
        cg.nodes[i].id = (int)i;
        if (nd.bus) {
            if (nd.type != "Tremolo::new") fail_unsupported("only Tremolo::new is available as a post-mix (bus) node in this version");
            if (out.bus_tremolo) fail_unsupported("only one post-mix (bus) node is supported in this version");
            out.bus_tremolo = true;
            out.channels = 2; // Frame<2>
            out.tremolo_rate = [](const UEnv&) { return 5.0f; };  // Tremolo::new() defaults, tremolo.rs:27-37
            out.tremolo_depth = [](const UEnv&) { return 0.5f; };
        } else if (!nti->emit) {
            fail("node type '" + nd.type + "' can only be used as a post-mix (bus) node");
        }
    }

    // ---- edges ------------------------------------------------------------------
    struct OutEdge {
        ExprP src;
        std::string policy;
    };
    std::map<int, std::vector<OutEdge>> out_edges; // graph output index -> sources
    std::vector<std::set<int>> deps(g.nodes.size()); // node -> nodes it reads
    std::vector<std::set<int>> out_deps(g.outputs.size());
    std::vector<std::set<int>> out_reads(g.outputs.size()); // output -> outputs its sources read (`out_a + out_b -> out`)
    std::map<int, std::pair<int, std::string>> ev_out_edges; // graph EVENT output -> (node, event-output port) feeding it
    std::vector<std::set<int>> fb_deps(g.nodes.size()); // feedback edges: liveness only, no ordering
    bool any_feedback = false;
    cg.emitted.assign(g.nodes.size(), 0);
    auto bus_node_of = [&](const std::string& endpoint) -> int {
        std::string nm = endpoint.substr(0, endpoint.find('.'));
        while (!nm.empty() && isspace((unsigned char)nm.back())) nm.pop_back();
        while (!nm.empty() && isspace((unsigned char)nm.front())) nm.erase(nm.begin());
        auto it = cg.node_by_name.find(nm);
        return (it != cg.node_by_name.end() && g.nodes[it->second].bus) ? it->second : -1;
    };
    int bus_src_output = -1, bus_final_output = -1;
    // ---- endpoint kinds, as far as the reference's macro knows them (ir/lower.rs:233-338 infer_endpoint_types): typed
    // graph inputs / outputs and stream-only policies seed kinds, connection statements propagate them both ways to a
    // fixpoint; node-to-node endpoints nobody typed stay unknown.  Their one consequence (codegen/emit_node.rs:35-58
    // classify_stream_fanin): SEVERAL edges into a destination known to be a VALUE are assigned one after the other --
    // the last one wins -- while a stream or unknown destination gets their sum.
    std::set<size_t> overwritten; // indices of g.edges a later edge into the same value destination replaces
    {
        enum K { Unknown = 0, Stream, Value, Event };
        auto strip = [](std::string t) {
            t.erase(std::remove_if(t.begin(), t.end(), [](char ch) { return isspace((unsigned char)ch); }), t.end());
            return t;
        };
        // "node.port", keyed by the ROOT node of an expanded array element (`voices__3.cutoff` -> `voices.cutoff`): the
        // reference infers one kind per (root node, field) and every element shares it (infer_endpoint_types runs before
        // the array is unrolled, ir/lower.rs:233-338) -- ADVICE r3
        auto canon = [](const std::string& key) {
            const size_t dot = key.find('.');
            std::string node = dot == std::string::npos ? key : key.substr(0, dot);
            const size_t us = node.rfind("__");
            if (us != std::string::npos && us + 2 < node.size() &&
                std::all_of(node.begin() + (long)us + 2, node.end(), [](char ch) { return isdigit((unsigned char)ch); }))
                node.erase(us);
            return dot == std::string::npos ? node : node + key.substr(dot);
        };
        std::map<std::string, K> known;
        auto decl_kind = [](Kind k) { return k == Kind::Stream ? Stream : (k == Kind::Value ? Value : Event); };
        std::function<K(const ExprP&)> kind_of = [&](const ExprP& e) -> K {
            if (!e) return Unknown;
            switch (e->t) {
            case Expr::Num: return Value;
            case Expr::Ref: {
                if (e->port.empty()) {
                    auto ii = cg.input_by_name.find(e->node);
                    if (ii != cg.input_by_name.end()) return decl_kind(g.inputs[ii->second].kind);
                    auto oi = cg.output_by_name.find(e->node);
                    return oi != cg.output_by_name.end() ? decl_kind(g.outputs[oi->second].kind) : Unknown;
                }
                auto it = known.find(canon(e->node + "." + e->port));
                return it == known.end() ? Unknown : it->second;
            }
            case Expr::Chan:
            case Expr::Neg: return kind_of(e->a);
            case Expr::Bin: {
                const K l = kind_of(e->a), r = kind_of(e->b);
                if (l == Unknown || r == Unknown || l == Event || r == Event) return Unknown;
                return (l == Stream || r == Stream) ? Stream : Value;
            }
            default: return Unknown; // calls and methods: no type inference for arbitrary functions
            }
        };
        struct Stmt {
            size_t idx;
            ExprP src;
            std::string dst, src_endpoint; // dst: "node.port" or an output name; src_endpoint: set when the source is a plain node endpoint
            bool dst_is_node;
        };
        std::vector<Stmt> stmts;
        for (size_t idx = 0; idx < g.edges.size(); ++idx) {
            const GEdge& e = g.edges[idx];
            if (bus_node_of(e.dst) >= 0 || bus_node_of(e.src) >= 0) continue;
            Stmt st;
            st.idx = idx;
            try {
                st.src = Parser(e.src).parse();
            } catch (const std::exception&) {
                continue; // (reported by the pass below)
            }
            st.dst = strip(e.dst);
            st.dst_is_node = st.dst.find('.') != std::string::npos;
            if (st.src->t == Expr::Ref && !st.src->port.empty()) st.src_endpoint = st.src->node + "." + st.src->port;
            if (e.policy == "linear" || e.policy == "sinc" || e.policy == "sinc_iir") { // stream-only kernels: both ends are streams
                if (st.dst_is_node) known.emplace(canon(st.dst), Stream);
                if (!st.src_endpoint.empty()) known.emplace(canon(st.src_endpoint), Stream);
            }
            stmts.push_back(st);
        }
        for (size_t round = 0; round <= stmts.size(); ++round) {
            bool changed = false;
            for (const Stmt& st : stmts) {
                const K sk = kind_of(st.src);
                if (sk != Unknown && st.dst_is_node && !known.count(canon(st.dst))) {
                    known[canon(st.dst)] = sk;
                    changed = true;
                }
                K dk = Unknown;
                if (st.dst_is_node) {
                    auto it = known.find(canon(st.dst));
                    dk = it == known.end() ? Unknown : it->second;
                } else {
                    auto oi = cg.output_by_name.find(st.dst);
                    if (oi != cg.output_by_name.end()) dk = decl_kind(g.outputs[oi->second].kind);
                }
                if (dk != Unknown && !st.src_endpoint.empty() && !known.count(canon(st.src_endpoint))) {
                    known[canon(st.src_endpoint)] = dk;
                    changed = true;
                }
            }
            if (!changed) break;
        }
        std::map<std::string, std::vector<size_t>> by_dst;
        for (const Stmt& st : stmts)
            if (!g.edges[st.idx].feedback) by_dst[st.dst].push_back(st.idx);
        for (const auto& kv : by_dst) {
            if (kv.second.size() < 2) continue;
            K dk = Unknown;
            if (kv.first.find('.') != std::string::npos) {
                auto it = known.find(canon(kv.first));
                dk = it == known.end() ? Unknown : it->second;
            } else {
                auto oi = cg.output_by_name.find(kv.first);
                if (oi != cg.output_by_name.end()) dk = decl_kind(g.outputs[oi->second].kind);
            }
            if (dk == Value)
                for (size_t k = 0; k + 1 < kv.second.size(); ++k) overwritten.insert(kv.second[k]);
        }
    }
    for (size_t edge_index = 0; edge_index < g.edges.size(); ++edge_index) {
        const GEdge& e = g.edges[edge_index];
        if (overwritten.count(edge_index)) continue; // (a later edge into this value destination replaces it)
        // ---- post-mix (bus) stage: `voices.output -> tremolo.input; v -> tremolo.depth; tremolo.output -> out`
        const int bdst = bus_node_of(e.dst), bsrc = bus_node_of(e.src);
        if (bdst >= 0 || bsrc >= 0) {
            if (bsrc >= 0) {
                auto oit = cg.output_by_name.find(e.dst);
                if (oit == cg.output_by_name.end()) fail("a bus node can only feed a graph output ('" + e.dst + "')");
                bus_final_output = oit->second;
                continue;
            }
            const std::string port = e.dst.substr(e.dst.find('.') + 1);
            if (port == "input") {
                auto oit = cg.output_by_name.find(e.src);
                if (oit == cg.output_by_name.end())
                    fail("bus node input must be fed by the voice graph's output name (the summed voices), got '" + e.src + "'");
                bus_src_output = oit->second;
            } else if (port == "rate" || port == "depth") {
                Val v = cg.eval(Parser(e.src).parse());
                if (!v.host) fail("bus node parameter '" + e.dst + "' must be a block-uniform value");
                if (v.rate == Rate::UFrame) // (the reference would move it every frame; the post-mix kernel takes one value per block)
                    fail("bus node parameter '" + e.dst + "' cannot be driven by a [ramp: N] input: it is applied once per block");
                (port == "rate" ? out.tremolo_rate : out.tremolo_depth) = v.host;
            } else {
                fail("bus node has no input '" + port + "'");
            }
            continue;
        }
        ExprP src = Parser(e.src).parse();
        std::vector<const Expr*> refs;
        collect_refs(src, refs);
        std::set<int> src_nodes, src_outputs;
        bool src_is_event_input = false, src_is_event_output = false;
        int src_event_input = -1;
        for (const Expr* r : refs) {
            if (r->port.empty()) {
                auto it = cg.input_by_name.find(r->node);
                if (it == cg.input_by_name.end()) {
                    auto oit = cg.output_by_name.find(r->node);
                    if (oit == cg.output_by_name.end()) fail("unknown source '" + r->node + "' in '" + e.src + "'");
                    src_outputs.insert(oit->second);
                    continue;
                }
                if (g.inputs[it->second].kind == Kind::Event) {
                    src_is_event_input = true;
                    src_event_input = out.inputs[it->second].event_index;
                }
            } else {
                auto it = cg.node_by_name.find(r->node);
                if (it == cg.node_by_name.end()) fail("unknown node '" + r->node + "' in '" + e.src + "'");
                src_nodes.insert(it->second);
                for (const std::string& eo : cg.nodes[it->second].type->ev_outputs)
                    if (eo == r->port) src_is_event_output = true;
            }
        }
        if (src_is_event_output && src->t != Expr::Ref) fail("an event output cannot be part of an expression ('" + e.src + "')");
        // destination
        std::string dn = e.dst, dp;
        size_t dot = dn.find('.');
        if (dot != std::string::npos) {
            dp = dn.substr(dot + 1);
            dn = dn.substr(0, dot);
            size_t par = dp.find('(');
            if (par != std::string::npos) dp = dp.substr(0, par);
        }
        auto trim = [](std::string& s) {
            while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back();
            while (!s.empty() && isspace((unsigned char)s.front())) s.erase(s.begin());
        };
        trim(dn);
        trim(dp);
        if (dp.empty()) {
            auto oit = cg.output_by_name.find(dn);
            if (oit == cg.output_by_name.end()) fail("unknown destination '" + e.dst + "'");
            if (g.outputs[oit->second].kind == Kind::Event) {
                // `node.trig -> x` with `output x: event;` (EventOutput, graph/types.rs:137-241): the events leave the
                // voice through the device log.  clear + copy: the LAST connected source delivers.
                if (!src_is_event_output || src->t != Expr::Ref)
                    fail("event output '" + e.dst + "' must be fed by an event output of a node ('" + e.src + "')");
                ev_out_edges[oit->second] = {*src_nodes.begin(), src->port};
                out_deps[oit->second].insert(src_nodes.begin(), src_nodes.end());
                continue;
            }
            if (src_is_event_input || src_is_event_output)
                fail("'" + e.dst + "' is a stream output: an event source cannot feed it");
            out_edges[oit->second].push_back({src, e.policy});
            out_deps[oit->second].insert(src_nodes.begin(), src_nodes.end());
            for (int so : src_outputs) {
                if (so == oit->second) fail("graph output '" + dn + "' reads itself");
                out_reads[oit->second].insert(so);
            }
            continue;
        }
        if (!src_outputs.empty())
            fail("a graph output can only be read by another graph output ('" + e.src + " -> " + e.dst + "')");
        auto nit = cg.node_by_name.find(dn);
        if (nit == cg.node_by_name.end()) fail("unknown destination node '" + dn + "'");
        NodeInst& dst = cg.nodes[nit->second];
        const PortSpec* ps = nullptr;
        for (const auto& p : dst.type->inputs)
            if (dp == p.name) ps = &p;
        if (!ps) fail("node '" + dn + "' (" + dst.decl->type + ") has no input '" + dp + "'");
        if (ps->kind == Kind::Event) {
            if (!(src_is_event_input || src_is_event_output) || src->t != Expr::Ref)
                fail("event input '" + e.dst + "' must be fed by a graph event input or by an event output of a node");
            // clear + copy: a later edge into the same input replaces the earlier source (last write wins)
            dst.ev_edges.erase(dp);
            dst.ev_node_edges.erase(dp);
            if (src_is_event_input) {
                dst.ev_edges[dp].push_back(src_event_input);
            } else {
                if (e.feedback) fail("an event edge cannot be a feedback edge ('" + e.src + "')");
                dst.ev_node_edges[dp] = {*src_nodes.begin(), src->port};
            }
        } else {
            if (src_is_event_input || src_is_event_output) fail("event source '" + e.src + "' cannot feed '" + e.dst + "'");
            dst.in_edges[dp].push_back({src, e.policy});
            if (e.feedback) { // outgoing leg of `src -> [via] -> dst` (ir/lower.rs:553-570)
                if (src->t != Expr::Ref || src->port.empty()) fail("a feedback edge must start at a node output ('" + e.src + "')");
                const int via = *src_nodes.begin();
                if (g.nodes[via].type.rfind("Delay::", 0) != 0)
                    fail("node '" + g.nodes[via].name + "' cannot close a feedback loop: only Delay implements AllowsFeedback");
                fb_deps[nit->second].insert(via);
                cg.fb_sources.insert("n" + std::to_string(via) + "." + src->port);
                any_feedback = true;
            } else {
                deps[nit->second].insert(src_nodes.begin(), src_nodes.end());
            }
        }
    }

    // a node that receives another node's events runs after it (and keeps it alive)
    for (size_t i = 0; i < g.nodes.size(); ++i)
        for (const auto& kv : cg.nodes[i].ev_node_edges) deps[i].insert(kv.second.first);

    // ---- dead-node removal (ir/passes/dead_nodes.rs:11-62) ------------------------
    if (!g.outputs.empty()) {
        std::vector<char> live(g.nodes.size(), 0);
        std::deque<int> q;
        for (auto& od : out_deps)
            for (int n : od) q.push_back(n);
        while (!q.empty()) {
            int n = q.front();
            q.pop_front();
            if (live[n]) continue;
            live[n] = 1;
            for (int d : deps[n]) q.push_back(d);
            for (int d : fb_deps[n]) q.push_back(d);
        }
        for (size_t i = 0; i < g.nodes.size(); ++i) cg.nodes[i].live = live[i];
    }
    for (size_t i = 0; i < g.nodes.size(); ++i)
        if (g.nodes[i].bus) cg.nodes[i].live = false;
    if (out.bus_tremolo && (bus_src_output < 0 || bus_final_output < 0))
        fail("the post-mix node must be wired `<voice output> -> node.input` and `node.output -> <graph output>`");
    for (size_t i = 0; i < g.nodes.size(); ++i)
        if (cg.nodes[i].live) out.lpv = std::max(out.lpv, cg.nodes[i].type->lpv);
    // harmonics per lane of an array-valued voice (OG_HPL).  Measured, electric piano, 262 144 voices (interleaved A/B,
    // profiles/r03_epiano_lanes.md): 16 lanes x 2 harmonics 0.95e11, 8 x 4 (round 2) 1.15e11, 4 x 8 1.26e11 at two
    // waves per SIMD and 1.28e11 at three -- the per-voice bookkeeping (ramp step, interpolation weights, event and
    // bus plumbing: ~15 of the 38 instructions a lane issues per frame at 8 x 4) is paid once per lane, so fewer lanes
    // per voice amortise it over more harmonics; 156 VGPRs still leave three waves per SIMD.  OGC_HPL=2|4|8 overrides;
    // the state layout [voice][32] does not depend on it.
    int hpl = 8;
    if (const char* eh = ogabi::experiment_knob("OGC_HPL")) {
        const int h = atoi(eh);
        if (h == 2 || h == 4 || h == 8) hpl = h;
    }
    if (out.lpv > 1) out.lpv = 32 / hpl;
    out.lane_width = out.lpv > 1 ? hpl : 1; // OG_HPL
    for (size_t i = 0; i < g.nodes.size(); ++i)
        if (cg.nodes[i].live && !cg.nodes[i].ev_node_edges.empty()) cg.dynamic_events = true;
    if (!ev_out_edges.empty()) cg.dynamic_events = true; // (the per-frame log / clear lives in the ordinary kernel's tick)
    // (array-valued graphs -- several lanes per voice: the per-voice scalars, event queues included, are replicated on the
    //  voice's lanes, every lane runs the handlers, and only the voice's lead lane logs a graph event output or counts a
    //  dropped push: og_kernel_rt.hip.h, VoiceCtx::lead)

    // ---- Kahn topological sort (ir/lower.rs:1015-1085), ready set in declaration order
    std::vector<int> order;
    {
        std::vector<int> indeg(g.nodes.size(), 0);
        std::vector<std::vector<int>> users(g.nodes.size());
        for (size_t i = 0; i < g.nodes.size(); ++i) {
            if (!cg.nodes[i].live) continue;
            for (int d : deps[i]) {
                if (d == (int)i) fail("node '" + g.nodes[i].name + "' feeds itself without a delay");
                indeg[i]++;
                users[d].push_back((int)i);
            }
        }
        std::deque<int> ready;
        size_t n_live = 0;
        for (size_t i = 0; i < g.nodes.size(); ++i)
            if (cg.nodes[i].live) {
                ++n_live;
                if (indeg[i] == 0) ready.push_back((int)i);
            }
        while (!ready.empty()) {
            int n = ready.front();
            ready.pop_front();
            order.push_back(n);
            for (int u : users[n])
                if (--indeg[u] == 0) ready.push_back(u);
        }
        if (order.size() != n_live)
            fail("graph contains a non-feedback cycle (use `-> [N] ->` to insert a delay buffer, or `-> [delay_node] ->` to "
                 "route through a declared Delay node)");
    }

    // ---- schedule order (round 4).  The Kahn order above is the reference's; it front-loads every source node
    // (FMVoice: the four envelopes, then the operators), which is the worst order for a pipeline cut: values live long,
    // and the heavy chain sits entirely behind the light sources.  When the graph is a pure per-frame dataflow -- no
    // feedback edge or delay (those read "the value of the previous frame", which depends on who runs first), no
    // node-to-node event edge (delivery order), one rate, one lane per voice -- ANY topological order computes the same
    // values bit for bit, so the nodes are emitted depth first from the sinks instead: a node's producers come right
    // before it, heaviest producer chain first, light leaves (envelopes) just ahead of their consumer:
    // FMVoice: env3 op3 route env2 op2 | mixer env1 op1 env_filter gain cutoff_mod filter output_gain -- the contiguous
    // two-wave cut now balances (56 | 54 estimated VALU per frame; the Kahn order could only offer 47 | 63), every
    // hand-off value but two stays inside its wave, and the ordinary kernel holds fewer values across nodes.
    // OGC_ALAP=0 keeps the Kahn order (A/B).
    {
        bool any_delay_node = false;
        for (int ni : order) any_delay_node = any_delay_node || cg.nodes[ni].decl->type.rfind("Delay::", 0) == 0;
        bool any_rate = false;
        for (int ni : order) any_rate = any_rate || cg.nodes[ni].decl->rate_factor != 1;
        const bool reorder = !(ogabi::experiment_knob("OGC_ALAP") && atoi(ogabi::experiment_knob("OGC_ALAP")) == 0) && !any_feedback && !cg.dynamic_events && !any_delay_node &&
                             !any_rate && out.lpv == 1 && order.size() >= 3;
        if (reorder) {
            std::vector<int> pos(g.nodes.size(), -1);
            for (size_t k = 0; k < order.size(); ++k) pos[order[k]] = (int)k;
            // weight of the not-yet-emitted producer cone of a node (memoised per call: the graphs are small)
            std::vector<char> done(g.nodes.size(), 0);
            std::function<int(int, std::set<int>&)> cone = [&](int n, std::set<int>& seen) -> int {
                if (done[n] || !seen.insert(n).second) return 0;
                int wsum = node_weight(cg.nodes[n].decl->type);
                for (int d : deps[n]) wsum += cone(d, seen);
                return wsum;
            };
            std::vector<int> sched;
            std::function<void(int)> visit = [&](int n) {
                if (done[n]) return;
                std::vector<std::pair<int, int>> ds; // (-cone weight, Kahn position): heaviest chain first, ties in Kahn order
                for (int d : deps[n]) {
                    if (done[d] || !cg.nodes[d].live) continue;
                    std::set<int> seen;
                    ds.push_back({-cone(d, seen), pos[d]});
                }
                std::sort(ds.begin(), ds.end());
                for (auto& pr : ds) visit(order[pr.second]);
                done[n] = 1;
                sched.push_back(n);
            };
            // sinks (nodes nobody reads) in Kahn order; what feeds the graph outputs comes last among them
            std::vector<int> n_users(g.nodes.size(), 0);
            for (int ni : order)
                for (int d : deps[ni]) n_users[d]++;
            for (int ni : order)
                if (n_users[ni] == 0) visit(ni);
            if (sched.size() == order.size()) {
                for (int ni : order) out.node_order.push_back(g.nodes[ni].name); // the reference's (Kahn) order, for the record
                order = sched;
            }
        }
    }

    // (feedback edges in graphs with oversampled nodes: allowed when both ends tick at the same rate -- checked where
    //  the consumer reads the previous tick's value, Codegen::eval)

    // ---- rate domains (emit_frame.rs:183-215; taint analysis emit_node.rs:516-584) ------------
    // 1 = oversampled inner loop; outer nodes downstream of an inner node run after the loop (2)
    for (int ni : order) {
        NodeInst& n = cg.nodes[ni];
        if (n.decl->rate_factor > 1) {
            n.domain = 1;
            for (int d : deps[ni])
                if (cg.nodes[d].domain == 2)
                    fail("oversampled node '" + n.decl->name + "' depends on '" + cg.nodes[d].decl->name +
                         "', which itself depends on the oversampled region");
        } else {
            n.domain = 0;
            for (int d : deps[ni])
                if (cg.nodes[d].domain >= 1) n.domain = 2;
        }
    }

    // ---- pipeline stages ------------------------------------------------------------------------
    // With one wave per SIMD (65 536 voices on 1024 SIMDs) a lone wave retires an instruction only
    // every ~6 cycles (dependent-instruction latency); more co-resident waves approach the ~3-cycle
    // measured issue ceiling.
    // The frame's node sequence is therefore also emitted as pipelines of two and of four waves over
    // the same 64 voices: every node (or run of envelopes) is a stage; a wave takes a contiguous group
    // of stages of balanced estimated VALU cost, works one hand-off chunk behind the previous wave,
    // and the values that cross between waves travel through LDS.  The ordinary kernel is the single
    // group of all stages.
    cg.stage_of.assign(g.nodes.size(), 0);
    cg.n_stages = 1;
    bool stereo_out = false;
    {
        const char* env_split = ogabi::experiment_knob("OGC_SPLIT");
        bool any_delay = false; // delay lines are staged per chunk by the ordinary kernel only
        for (int ni : order) any_delay = any_delay || cg.nodes[ni].decl->type.rfind("Delay::", 0) == 0;
        // a Frame<2> voice output (two bus tiles) is summed by the ordinary kernel only
        std::function<int(const ExprP&)> width_of = [&](const ExprP& e) -> int {
            if (!e) return 1;
            if (e->t == Expr::Ref && !e->port.empty()) {
                auto nit = cg.node_by_name.find(e->node);
                if (nit == cg.node_by_name.end()) return 1;
                const NodeTypeInfo* ti = cg.nodes[nit->second].type;
                for (size_t o = 0; o < ti->outputs.size() && o < ti->out_channels.size(); ++o)
                    if (e->port == ti->outputs[o]) return ti->out_channels[o];
                return 1;
            }
            if (e->t == Expr::Ref) { // a Frame<N> stream input of the graph
                auto iit = cg.input_by_name.find(e->node);
                return iit == cg.input_by_name.end() ? 1 : std::max(1, out.inputs[iit->second].decl.channels);
            }
            if (e->t == Expr::Chan || e->t == Expr::Method) return 1;
            if (e->t == Expr::Call) {
                if (e->port == "Frame") return (int)e->args.size();
                const UserFunction* f = lookup_function(e->node, e->port);
                return f ? f->result_channels : 1;
            }
            return std::max(width_of(e->a), width_of(e->b));
        };
        for (auto& kv : out_edges)
            for (auto& src : kv.second) stereo_out = stereo_out || width_of(src.src) >= 2;
        for (size_t oi = 0; oi < g.outputs.size(); ++oi) // (a declared `output out: stream: Frame<2>`)
            stereo_out = stereo_out || (g.outputs[oi].kind == Kind::Stream && g.outputs[oi].channels >= 2);
        // (several stream outputs = several bus channels: summed by the ordinary kernel only, like a Frame<2> output)
        {
            int n_stream_outs = 0;
            for (size_t oi = 0; oi < g.outputs.size(); ++oi)
                if (g.outputs[oi].kind == Kind::Stream && out_edges.count((int)oi) && (int)oi != bus_src_output) ++n_stream_outs;
            if (n_stream_outs > 1 && !out.bus_tremolo) stereo_out = true;
        }
        bool want = !(env_split && atoi(env_split) == 0) && cg.N == 1 && out.lpv == 1 && order.size() >= 2 && !any_feedback && !cg.dynamic_events &&
                    !any_delay; // (round 4: several bus channels -- Frame<N> / several stream outputs -- run in the pipelines too)
        (void)stereo_out;
        // estimated per-tick VALU cost of every node, in emission order
        int total = 0;
        std::vector<int> w;
        for (int ni : order) {
            const NodeInst& n = cg.nodes[ni];
            int wt = node_weight(n.decl->type);
            if (n.decl->type.rfind("TptFilter", 0) == 0) { // a cutoff fed by another node moves every sample: the
                bool moving = false;                         // coefficient update (tan, reciprocal) runs on every tick
                for (const char* port : {"cutoff", "q", "f_mod"}) {
                    auto it = n.in_edges.find(port);
                    if (it == n.in_edges.end()) continue;
                    for (const auto& src : it->second) {
                        std::vector<const Expr*> refs;
                        collect_refs(src.e, refs);
                        for (const Expr* r : refs) moving = moving || !r->port.empty();
                    }
                }
                // (its true cost is ~45, but the grouping this weight gives for fm_voice -- op2 with the consumer wave --
                //  measured equal on the default run and 12% faster at 98 304 voices than op2 with the producer)
                wt = moving ? 12 : 9; // core 7 + the watched-input test + the coefficient update on the frames whose cutoff moved
            }
            w.push_back(wt);
            total += wt;
        }
        // Stage = unit: a node, or a run of consecutive envelopes (their stage-end check is shared, see
        // flush_post()).  At most 16 stages: the lightest adjacent pair is merged until they fit.
        std::vector<int> unit_w, unit_end; // weight and one-past-last order index of every unit
        for (size_t k = 0; k < order.size(); ++k) {
            const bool env = cg.nodes[order[k]].decl->type.rfind("AdsrEnvelope::", 0) == 0;
            const bool prev_env = k > 0 && cg.nodes[order[k - 1]].decl->type.rfind("AdsrEnvelope::", 0) == 0;
            if (env && prev_env) {
                unit_w.back() += w[k];
                unit_end.back() = (int)k + 1;
            } else {
                unit_w.push_back(w[k]);
                unit_end.push_back((int)k + 1);
            }
        }
        while (unit_w.size() > 16) {
            size_t best = 0;
            for (size_t k = 0; k + 1 < unit_w.size(); ++k)
                if (unit_w[k] + unit_w[k + 1] < unit_w[best] + unit_w[best + 1]) best = k;
            unit_w[best] += unit_w[best + 1];
            unit_end[best] = unit_end[best + 1];
            unit_w.erase(unit_w.begin() + best + 1);
            unit_end.erase(unit_end.begin() + best + 1);
        }
        // contiguous grouping of the stages into `parts` waves minimising the heaviest wave (the last wave
        // also carries the mix-bus work); ties: smallest sum of squares
        // (the wave that closes the chunk also transposes and reduces the bus tile and is the one everybody waits for:
        //  interleaved A/B of the cuts, round 5 -- gain + add with the wave in front of the filter wave +1.5 .. +3.7 %)
        const int BUS_W = 8;
        // Values that cross a cut.  A value produced in one wave and read in another costs the producer an LDS store per
        // frame, every reading wave an LDS load per frame AND eight registers (a chunk's hand-off values are fetched
        // before its first tick): the cut {env2 op2 mixer env1}|{op1 env_filter gain add} balances the instruction counts
        // better than {env2 op2 mixer}|{env1 op1 env_filter gain add} but reads one more value in its heaviest wave -- 92
        // VGPRs instead of 80, five resident workgroups per CU instead of six, and measured 1.5 % slower
        // (profiles/r05g_session6.log).  unit_of[node] = stage of every scheduled node; refs by (node, port).
        std::map<std::string, int> unit_of_name;
        {
            int st = 0;
            for (int k = 0; k < (int)order.size(); ++k) {
                while (k >= unit_end[st]) ++st;
                unit_of_name[cg.nodes[order[k]].decl->name] = st;
            }
        }
        std::vector<std::set<std::pair<int, std::string>>> reads_of((size_t)unit_w.size()); // per stage: (producer stage, "node.port") it reads
        for (int k = 0; k < (int)order.size(); ++k) {
            const NodeInst& nd = cg.nodes[order[k]];
            const int me = unit_of_name[nd.decl->name];
            for (const auto& kv : nd.in_edges)
                for (const auto& src : kv.second) {
                    std::vector<const Expr*> refs;
                    collect_refs(src.e, refs);
                    for (const Expr* r : refs) {
                        auto it = unit_of_name.find(r->node);
                        if (r->port.empty() || it == unit_of_name.end() || it->second == me) continue;
                        reads_of[(size_t)me].insert({it->second, r->node + "." + r->port});
                    }
                }
        }
        std::set<std::pair<int, std::string>> bus_reads; // what the graph outputs read: consumed by the LAST wave
        for (auto& kv : out_edges)
            for (auto& src : kv.second) {
                std::vector<const Expr*> refs;
                collect_refs(src.src, refs);
                for (const Expr* r : refs) {
                    auto it = unit_of_name.find(r->node);
                    if (!r->port.empty() && it != unit_of_name.end()) bus_reads.insert({it->second, r->node + "." + r->port});
                }
            }
        auto crossing = [&](int b0, int e, bool is_last) { // extra weight of the wave holding stages [b0, e)
            std::set<std::string> in, outv;
            for (int u = b0; u < e; ++u)
                for (const auto& rd : reads_of[(size_t)u])
                    if (rd.first < b0 || rd.first >= e) in.insert(rd.second);
            if (is_last)
                for (const auto& rd : bus_reads)
                    if (rd.first < b0 || rd.first >= e) in.insert(rd.second);
            for (int u = 0; u < (int)unit_w.size(); ++u) {
                if (u >= b0 && u < e) continue;
                for (const auto& rd : reads_of[(size_t)u])
                    if (rd.first >= b0 && rd.first < e) outv.insert(rd.second);
            }
            if (!is_last)
                for (const auto& rd : bus_reads)
                    if (rd.first >= b0 && rd.first < e) outv.insert(rd.second);
            // (3 per value read -- load, wait and eight registers --, 1 per value written: with these the DP picks the cut
            //  that measured best of six, {env3 op3 xf}|{env2 op2 mix}|{env1 op1 env_f gain add}|{filter gain bus};
            //  2 per read picked {env3 op3 xf env2}|{op2 mix env1}|..., which measured worst)
            return 3 * (int)in.size() + (int)outv.size();
        };
        auto grouping = [&](int parts) {
            const int n = (int)unit_w.size();
            std::vector<int> pre(n + 1, 0);
            for (int k = 0; k < n; ++k) pre[k + 1] = pre[k] + unit_w[k];
            struct Best {
                long mx = 1L << 40, sq = 1L << 40;
                std::vector<int> ends; // one-past-last stage of every wave
            };
            std::vector<std::vector<Best>> dp(parts + 1, std::vector<Best>(n + 1));
            dp[0][0].mx = 0;
            dp[0][0].sq = 0;
            for (int p2 = 1; p2 <= parts; ++p2)
                for (int e = p2; e <= n; ++e)
                    for (int b0 = p2 - 1; b0 < e; ++b0) {
                        const Best& prev = dp[p2 - 1][b0];
                        if (prev.mx >= (1L << 40)) continue;
                        const bool is_last = p2 == parts && e == n;
                        const long wgt = pre[e] - pre[b0] + (is_last ? BUS_W : 0) + crossing(b0, e, is_last);
                        const long mx = std::max(prev.mx, wgt), sq = prev.sq + wgt * wgt;
                        Best& cur = dp[p2][e];
                        if (mx < cur.mx || (mx == cur.mx && sq < cur.sq)) {
                            cur.mx = mx;
                            cur.sq = sq;
                            cur.ends = prev.ends;
                            cur.ends.push_back(e);
                        }
                    }
            std::vector<std::vector<int>> groups;
            int lo = 0;
            for (int e : dp[parts][n].ends) {
                groups.emplace_back();
                for (int k = lo; k < e; ++k) groups.back().push_back(k);
                lo = e;
            }
            return groups;
        };
        if (want && total >= 34 && unit_w.size() >= 2) { // (thresholds in units of the round-5 node weights)
            cg.n_stages = (int)unit_w.size();
            int st = 0;
            for (int k = 0; k < (int)order.size(); ++k) {
                while (k >= unit_end[st]) ++st;
                cg.stage_of[order[k]] = st;
            }
            cg.groups2 = grouping(2);
            if (const char* ec = ogabi::experiment_knob("OGC_CUT2")) { // experiment knob: first wave = the first k stages
                const int ku = std::max(1, std::min(cg.n_stages - 1, atoi(ec)));
                cg.groups2 = {{}, {}};
                for (int k = 0; k < cg.n_stages; ++k) cg.groups2[k < ku ? 0 : 1].push_back(k);
            }
            const char* ep = ogabi::experiment_knob("OGC_PARTS");
            if (total >= 48 && unit_w.size() >= 4 && !(ep && atoi(ep) < 4)) cg.groups4 = grouping(ogabi::experiment_knob("OGC_K3") ? 3 : 4);
            if (const char* ec = ogabi::experiment_knob("OGC_CUTS")) { // experiment knob: "a,b[,c]" = one-past-last stage of every wave but the last (3 or 4 waves)
                std::vector<int> ends;
                for (const char* q = ec; *q;) {
                    ends.push_back(atoi(q));
                    while (*q && *q != ',') ++q;
                    if (*q == ',') ++q;
                }
                ends.push_back(cg.n_stages);
                std::vector<std::vector<int>> gr;
                int lo = 0;
                bool ok = ends.size() >= 3 && ends.size() <= 6;
                for (int e : ends) {
                    ok = ok && e > lo && e <= cg.n_stages;
                    gr.emplace_back();
                    for (int k = lo; k < e && ok; ++k) gr.back().push_back(k);
                    lo = e;
                }
                if (ok) cg.groups4 = gr;
            }
        }
    }
    cg.split = cg.n_stages > 1;


    // ---- event outputs read inside the oversampled region (step 6a, see Sect::s_ev6a): which they are, and the queue
    // capacity of the graph BEFORE any node is emitted (such a queue collects the pushes of all N inner ticks of a frame
    // and is a state plane per slot)
    for (int ni : order) {
        const NodeInst& n = cg.nodes[ni];
        if (n.type->user && !n.type->user->ev_outputs.empty() && n.type->user->event_capacity > cg.ev_capacity)
            cg.ev_capacity = std::min(32, n.type->user->event_capacity);
    }
    for (int ni : order) {
        const NodeInst& n = cg.nodes[ni];
        if (n.domain == 0) continue;
        for (const auto& kv : n.ev_node_edges) {
            const NodeInst& src = cg.nodes[kv.second.first];
            if (src.domain != 1) continue;
            (n.domain == 1 ? cg.inner_ev_sources : cg.outer_ev_sources).insert({src.id, kv.second.second});
            const int per_tick = src.type->user && src.type->user->event_capacity > 0 ? src.type->user->event_capacity : 2;
            cg.ev_capacity = std::max(cg.ev_capacity, std::min(32, per_tick * cg.N)); // the reference's queue holds 32
            for (const auto& eo : ev_out_edges)
                if (eo.second.first == src.id && eo.second.second == kv.second.second)
                    fail_unsupported("event output '" + src.decl->name + "." + kv.second.second + "' feeds both an oversampled node and a graph event output");
        }
    }

    // ---- emit nodes: outer (pre), inner, outer (post), each in topological order -------------
    for (int dom = 0; dom < 3; ++dom) {
        cg.dom = dom;
        for (int ni : order) {
            NodeInst& n = cg.nodes[ni];
            if (n.domain != dom) continue;
            const bool is_env = n.decl->type.rfind("AdsrEnvelope::", 0) == 0;
            if (!is_env || cg.cs != cg.stage_of[ni]) cg.flush_post(); // (into the stream of the stage that owns them)
            cg.cs = cg.stage_of[ni];
            NodeCtx x{cg, n, "n" + std::to_string(n.id) + "_"};
            cg.os() << "        // " << n.decl->name << " = " << n.decl->type
                    << (dom == 1 ? " * " + std::to_string(cg.N) : std::string()) << "\n";
            n.type->emit(x);
            cg.emitted[ni] = 1;
            out.schedule_order.push_back(n.decl->name);
        }
        cg.flush_post();
    }

    // ---- graph output (outer rate, last stage) -------------------------------------------------
    std::string bus_expr = "0.0f";
    cg.dom = 2;
    cg.cs = cg.n_stages - 1;
    {
        int n_stream = 0;
        // outputs that other outputs read come first (`[sinc] a.output -> out_a; out_a + out_b -> out`); such an output
        // is a named per-frame value of the voice, the output nobody reads is the one that goes onto the mix bus
        std::vector<int> consumed(g.outputs.size(), 0);
        for (const auto& rd : out_reads)
            for (int so : rd) consumed[so] = 1;
        std::vector<size_t> oorder;
        {
            std::vector<int> done(g.outputs.size(), 0);
            for (size_t pass = 0; pass <= g.outputs.size() && oorder.size() < g.outputs.size(); ++pass)
                for (size_t oi = 0; oi < g.outputs.size(); ++oi) {
                    if (done[oi]) continue;
                    bool ready = true;
                    for (int so : out_reads[oi]) ready = ready && done[so];
                    if (!ready) continue;
                    done[oi] = 1;
                    oorder.push_back(oi);
                }
            if (oorder.size() < g.outputs.size()) fail("graph outputs read each other in a cycle");
        }
        for (size_t oi = 0; oi < g.outputs.size(); ++oi) { // event outputs: one log call per frame each, before the clears
            if (g.outputs[oi].kind != Kind::Event) continue;
            auto eo = ev_out_edges.find((int)oi);
            if (eo == ev_out_edges.end()) continue; // declared, never fed: nothing ever arrives
            const NodeInst& src = cg.nodes[eo->second.first];
            const std::string q = "n" + std::to_string(src.id) + "_" + eo->second.second;
            // an oversampled node's queue is cleared after every INNER tick: its events are logged there, all N ticks of
            // outer frame f under frame f (the inner -> outer rescale of a cross-rate event edge, ir/lower.rs:846-852)
            std::ostringstream& log = src.domain == 1 ? cg.sec[cg.stage_of.size() > (size_t)src.id ? cg.stage_of[src.id] : 0].s_log : cg.frame_log;
            log << (src.domain == 1 ? "    " : "") << "        if (__any((int)(" << q << ".n != 0u))) og::ev_out_log(A, c, " << out.event_outputs.size() << "u, f, " << q << ");\n";
            out.event_outputs.push_back(g.outputs[oi].name);
        }
        std::vector<int> formed(g.outputs.size(), 0); // channels of every stream output that has sources
        for (size_t oi : oorder) {
            if (g.outputs[oi].kind == Kind::Event) continue;
            auto it = out_edges.find((int)oi);
            if (it == out_edges.end()) {
                if (consumed[oi]) fail("graph output '" + g.outputs[oi].name + "' is read but nothing feeds it");
                continue;
            }
            if ((int)oi == bus_final_output) fail("graph output fed by the post-mix node cannot have other sources");
            ++n_stream;
            // a Frame<N> voice (N = 2..4): every channel summed over the voices, bus interleaved (BlockRender<Frame<N>>)
            std::vector<std::string> acc; // one accumulated expression per channel
            int width = 0;                // 1 = f32, N = Frame<N>
            for (size_t k = 0; k < it->second.size(); ++k) {
                if (it->second.size() > 1 && !it->second[k].policy.empty())
                    fail("fan-in summing supports only same-rate sources (graph output)");
                Val v = cg.cross(cg.eval(it->second[k].src), it->second[k].policy, false, false);
                const int w = v.is_frame() ? (int)v.ch.size() : 1;
                if (w > 4) fail("graph output '" + g.outputs[oi].name + "' is fed a Frame<" + std::to_string(w) + ">: the mix bus carries f32 or Frame<2..4> voices");
                if (k > 0 && w != width)
                    fail("graph output '" + g.outputs[oi].name + "' mixes " + (width > 1 ? "Frame<" + std::to_string(width) + ">" : std::string("f32")) +
                         " and " + (w > 1 ? "Frame<" + std::to_string(w) + ">" : std::string("f32")) + " sources");
                width = w;
                if (w > 1) {
                    if (out.bus_tremolo) // (array-valued voices: the voice's lead lane puts every channel, og::bus_put)
                        fail_unsupported("a Frame<N> graph output is not supported in post-mix graphs yet");
                    if (k == 0) acc.assign((size_t)w, std::string());
                    for (int c = 0; c < w; ++c) acc[(size_t)c] = (k == 0) ? v.ch[(size_t)c].e : "(" + acc[(size_t)c] + " + " + v.ch[(size_t)c].e + ")";
                    continue;
                }
                if (it->second.size() > 1 && v.inner) fail("fan-in summing supports only same-rate sources (graph output)");
                if (k == 0) acc.assign(1, std::string());
                acc[0] = (k == 0) ? v.e : "(" + acc[0] + " + " + v.e + ")";
            }
            const int declared = g.outputs[oi].channels; // `output out: stream: Frame<2>;` (0: not declared)
            if (declared > 4) fail("graph output '" + g.outputs[oi].name + "': the mix bus carries f32 or Frame<2..4> voices");
            if (declared && declared != width)
                fail("graph output '" + g.outputs[oi].name + "' is declared " + (declared > 1 ? "Frame<" + std::to_string(declared) + ">" : std::string("f32")) +
                     " but fed " + (width > 1 ? "a Frame<" + std::to_string(width) + ">" : std::string("an f32 stream")));
            // every stream output is a named value of this frame: other outputs may read it (`out_a + out_b -> out`),
            // and ALL of them go onto the mix bus, one channel (Frame<N>: N) each, in declaration order
            const std::string var = "go" + std::to_string(oi);
            auto chan_name = [&](int c) { // (Frame<2> keeps its _l / _r names: the text of existing kernels does not move)
                return width == 2 ? var + (c == 0 ? "_l" : "_r") : var + "_c" + std::to_string(c);
            };
            Val ov;
            ov.rate = Rate::Vary;
            if (width > 1) {
                cg.os() << "        const float ";
                for (int c = 0; c < width; ++c) {
                    cg.os() << (c ? ", " : "") << chan_name(c) << " = " << acc[(size_t)c];
                    Val cv;
                    cv.rate = Rate::Vary;
                    cv.e = chan_name(c);
                    ov.ch.push_back(cv);
                }
                cg.os() << ";\n";
            } else {
                cg.os() << "        const float " << var << " = " << acc[0] << ";\n";
                ov.e = var;
            }
            cg.output_vals[g.outputs[oi].name] = ov;
            formed[oi] = width;
        }
        // the bus: the formed stream outputs in DECLARATION order
        std::vector<std::string> chans;
        for (size_t oi = 0; oi < g.outputs.size(); ++oi) {
            if (!formed[oi]) continue;
            const std::string var = "go" + std::to_string(oi);
            out.output_channels.push_back({g.outputs[oi].name, (int)chans.size(), formed[oi]});
            if (formed[oi] == 2) {
                chans.push_back(var + "_l");
                chans.push_back(var + "_r");
            } else if (formed[oi] > 2) {
                for (int c = 0; c < formed[oi]; ++c) chans.push_back(var + "_c" + std::to_string(c));
            } else {
                chans.push_back(var);
            }
        }
        if (chans.size() > 1 && out.bus_tremolo)
            fail_unsupported("several bus channels (stream outputs / Frame<2>) are not supported in post-mix graphs yet");
        if (chans.size() > 4) fail("the mix bus carries at most 4 channels (stream outputs, a Frame<N> counting N)");
        if (chans.size() == 1) {
            cg.os() << "        const float g_out = " << chans[0] << ";\n";
            bus_expr = "g_out";
        } else if (chans.size() > 1) {
            cg.os() << "        const og::OutN<" << chans.size() << "> g_out = {{";
            for (size_t k = 0; k < chans.size(); ++k) cg.os() << (k ? ", " : "") << chans[k];
            cg.os() << "}};\n";
            out.voice_channels = (uint32_t)chans.size();
            out.channels = (uint32_t)chans.size();
            bus_expr = "g_out";
        }
        (void)n_stream;
    }
    out.can_split = cg.split;
    out.max_pipeline = !cg.groups4.empty() ? 4 : (!cg.groups2.empty() ? 2 : 1);

    if (out.n_slots > 160) fail("graph needs more than 160 uniform slots");

    // per-stage pieces of the kernel body
    auto tick_code = [&](int st) {
        Codegen::Sect& S = cg.sec[st];
        std::ostringstream t;
        t << S.s_pre.str() << S.s_up.str() << S.s_ev6a.str() << S.s_ev6a_clear.str();
        if (cg.N > 1)
            t << "#pragma unroll\n        for (int j = 0; j < " << cg.N << "; ++j) { // oversampled inner loop\n"
              << S.s_inner.str() << S.s_log.str() << S.s_cap.str() << "        }\n";
        t << S.s_down.str() << S.s_post.str();
        return t.str();
    };
    // per-voice events due on frame f (sub-block splitting of process_block, codegen/mod.rs:836-871)
    auto events_code = [&](const std::vector<int>& stages, bool prefetch = true) {
        const std::string PRE = prefetch ? "true" : "false"; // (the ordinary kernel reads the record when it fires: og::ev_arm)
        std::ostringstream t;
        t << "    auto events = [&](const uint32_t f) __attribute__((always_inline)) {\n"
          << "        if (f == c.next_ev) {\n"
          << "            do {\n"
          << "                const og::EvRec ev = og::ev_record<" << PRE << ">(A, c);\n";
        bool first = true;
        for (size_t i = 0; i < out.inputs.size(); ++i) {
            const InputInfo& in = out.inputs[i];
            if (in.decl.kind == Kind::Value && in.decl.per_voice) {
                t << "                " << (first ? "" : "else ") << "if (ev.target == (OG_EV_SETVALUE | " << i
                  << "u)) { vin_" << i << " = ev.value; derive(); }\n";
                first = false;
            }
        }
        std::map<int, std::string> handlers;
        for (int st : stages)
            for (auto& kv : cg.sec[st].ev_handlers) handlers[kv.first] += kv.second.str();
        for (auto& kv : handlers) {
            t << "                " << (first ? "" : "else ") << "if (ev.target == " << kv.first << "u) {\n" << kv.second
              << "                }\n";
            first = false;
        }
        t << "                og::ev_advance<" << PRE << ">(A, c);\n"
          << "            } while (c.next_ev <= f);\n"
          << "        }\n    };\n";
        return t.str();
    };
    auto vin_store = [&]() {
        std::ostringstream t;
        for (size_t i = 0; i < out.inputs.size(); ++i) {
            const InputInfo& in = out.inputs[i];
            if (in.decl.kind == Kind::Value && in.decl.per_voice)
                t << "        if (c.ev_cur != c.ev_cur0) og::st_f(A, c, " << in.state_word << ", vin_" << i << ");\n";
        }
        return t.str();
    };
    // frames per straight-line scheduling region of the ordinary kernel's quiet-chunk loop.  A large graph with
    // envelopes has three copies of the chunk body (no stage-end checks / no release arithmetic) and sits at the
    // 128-VGPR cap: unrolling by two spills ~20 registers there (fm_voice: 2-4% slower at >= 131 072 voices, where
    // this kernel runs); small graphs gain from it (4x saturator +5%).
    int graph_weight = 0;
    bool has_env = false;
    for (int ni : order) {
        graph_weight += node_weight(cg.nodes[ni].decl->type);
        has_env = has_env || cg.nodes[ni].decl->type.rfind("AdsrEnvelope::", 0) == 0;
    }
    out.valu_estimate = graph_weight + 8; // + the mix bus
    int unroll = (has_env && graph_weight >= 100) ? 1 : 2;
    // Round 6: the whole 16-frame bus chunk as ONE straight-line region wherever registers allow.  A value that lives across
    // frames -- the saturator's four half-band histories (62 live samples that shift every frame), phases, filter states --
    // crosses the loop's back edge once per trip and has to sit where the loop head expects it: with two frames per trip the
    // histories were copied on every trip (20 v_mov per frame of 162 VALU); with sixteen a sample is written straight into the
    // register it will be read from a chunk later.  Interleaved A/B on one MI355X (profiles/r06_handoff_ab.md, r07f / r07g):
    // SatGraph_4x 131 072 voices 3.33e11 -> 3.50 / 3.61 / 3.71e11 at 4 / 8 / 16; fm_voice 262 144 6.88e11 -> 6.97 / 6.92 /
    // 7.02e11, 1 048 576 8.06e11 -> 8.17 / 8.06 / 8.17e11; osc+env+TPT 1.125e12 -> 1.157 / 1.161 / 1.153e12; the delay-line
    // voice 5.21e11 -> 5.32 / 5.28 / 5.17e11 (its chunk is bracketed by the ring staging: four frames per trip); the
    // electric piano does not move (its harmonics are not shifted) and keeps its two.  (The pipelined shapes always
    // unrolled their chunks.)
    if (unroll > 1) {
        if (out.lpv > 1) unroll = 2;
        else if (!out.rings.empty()) unroll = 4;
        else if (cg.N > 1) unroll = 16;
        else unroll = (has_env && graph_weight >= 70) ? 4 : 8; // (fm_voice: sixteen is worth +0.7 % over four and brings spills into the tapped variants' chunk bodies)
    }
    if (const char* u = ogabi::experiment_knob("OGC_UNROLL")) unroll = std::max(1, std::min(16, atoi(u)));

    // A kernel is assembled from GROUPS of consecutive stages, one wave per group: the ordinary kernel
    // has the single group {0..n-1}, the pipelined ones two or four groups.
    const int NS = cg.n_stages;
    auto cat = [&](const std::vector<int>& st, std::ostringstream Codegen::Sect::*m) {
        std::string r;
        for (int k : st) r += (cg.sec[k].*m).str();
        return r;
    };
    // group index of a stage under a grouping
    auto group_of = [](const std::vector<std::vector<int>>& groups, int stage) {
        for (size_t gi = 0; gi < groups.size(); ++gi)
            for (int k : groups[gi])
                if (k == stage) return (int)gi;
        return -1;
    };
    // per-frame code of one group: stage code in order; a value read by a later stage is aliased right
    // after its producer (same wave) or travels through its LDS channel (other wave)
    auto group_tick = [&](const std::vector<std::vector<int>>& groups, int gi) {
        std::ostringstream t;
        const std::vector<int>& st = groups[gi];
        for (size_t k = 0; k < cg.xvals.size(); ++k) {
            const auto& xv = cg.xvals[k];
            bool used_here = false;
            for (int u : xv.users) used_here = used_here || group_of(groups, u) == gi;
            if (used_here && group_of(groups, xv.from) != gi) // (quiet chunks: read at the top of the chunk into xp<k>[])
                t << "        const float " << xv.alias << " = [&]() __attribute__((always_inline)) { if constexpr (decltype(chk)::pre) return xp"
                  << k << "[j]; else return chan" << k << "[ch % XD" << k << "][j][c.lane]; }();\n";
        }
        for (int fs : st) {
            t << tick_code(fs);
            for (size_t k = 0; k < cg.xvals.size(); ++k) {
                const auto& xv = cg.xvals[k];
                if (xv.from != fs) continue;
                bool local = false, remote = false;
                for (int u : xv.users) (group_of(groups, u) == gi ? local : remote) = true;
                if (local) t << "        const float " << xv.alias << " = " << xv.var << ";\n";
                if (remote) t << "        chan" << k << "[ch % XD" << k << "][j][c.lane] = " << xv.var << ";\n";
            }
        }
        return t.str();
    };
    std::vector<int> all_stages;
    for (int k = 0; k < NS; ++k) all_stages.push_back(k);

    // ---- ordinary kernel: one wave = 64 voices, the whole node sequence -------------------------
    std::ostringstream body;
    body << "template <bool RAMPS, bool TAPS>\n"
         << "__device__ __forceinline__ void voice_block(const OgBlockArgs& A)\n{\n"
         << "    __shared__ og::" << (out.voice_channels > 1 ? "BusLdsN<" + std::to_string(out.voice_channels) + ">" : std::string("BusLds")) << " bus;\n"
         << (out.rings.empty() ? std::string()
                               : "    __shared__ float ring_lds[" + std::to_string(out.rings.size()) +
                                     "][OG_BUS_CHUNK][OG_WAVE];\n    uint32_t cbase = 0;\n")
         << "    og::VoiceCtx c;\n"
         << "    og::voice_begin<TAPS, LPV>(A, c);\n"
         << "    og::bus_init(c, bus);\n"
         << (ogabi::experiment_knob("OGC_PRIO_PARITY") ? "    if ((blockIdx.x >> 3) & 1u) __builtin_amdgcn_s_setprio(1); // experiment\n" : "")
         << cg.common_decl.str() << cat(all_stages, &Codegen::Sect::decl) << "    if (c.valid) {\n"
         << cg.common_load.str() << cat(all_stages, &Codegen::Sect::load) << "    }\n";
    body << "    auto derive = [&]() __attribute__((always_inline)) {\n" << cat(all_stages, &Codegen::Sect::derive) << "    };\n";
    body << cat(all_stages, &Codegen::Sect::pre) << "    derive();\n";
    if (!out.rings.empty()) // state loads complete here, so that no wait for them lands inside the chunk loop,
                            // where it would also sit out the delay-line loads staged for the next chunk
        body << "    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)\n";
    // one frame of the voice graph (nodes in topological order); returns the voice's output sample
    // min over the countdowns of a group's envelopes, or "" when it has none
    // (dynamic_events: a gate from another node can start a Release or a new stage on any frame)
    const bool chunk_chk = !(ogabi::experiment_knob("OGC_CHUNK_CHK") && atoi(ogabi::experiment_knob("OGC_CHUNK_CHK")) == 0) && !cg.dynamic_events;
    auto min_cnt = [&](const std::vector<int>& st) {
        std::string m;
        if (!chunk_chk) return m;
        for (int k : st)
            for (const auto& cexp : cg.sec[k].env_cnts) m = m.empty() ? cexp : "min(" + m + ", " + cexp + ")";
        return m;
    };
    auto rs_sum = [&](const std::vector<int>& st) {
        std::string m;
        for (int k : st)
            for (const auto& r : cg.sec[k].env_rs) m = m.empty() ? r : "(" + m + " + " + r + ")";
        return m;
    };
    // `<E>.fc = (float)<E>.cnt;` for the envelopes of a group: in front of every loop whose ticks run the release arithmetic
    // (the release-free chunk variant steps cnt only), see og::Adsr::fc
    auto fc_sync = [&](const std::vector<int>& st, const std::string& ind) {
        std::string r;
        for (int k : st)
            for (const auto& cexp : cg.sec[k].env_cnts) {
                const std::string E = cexp.substr(0, cexp.size() - 4); // "<E>.cnt"
                r += ind + E + ".fc = (float)" + E + ".cnt;\n";
            }
        return r;
    };
    body << "    auto tick = [&](const uint32_t f, auto chk) __attribute__((always_inline)) -> " << (out.voice_channels > 1 ? "og::OutN<" + std::to_string(out.voice_channels) + ">" : std::string("float")) << " {\n"
         << group_tick({all_stages}, 0);
    if (cg.frame_end.str().empty()) {
        body << "        return " << bus_expr << ";\n    };\n";
    } else { // clear_event_outputs(): the frame's node-to-node events have been delivered (and the graph's event outputs logged)
        body << "        const auto g_bus = " << bus_expr << ";\n" << cg.frame_log.str() << cg.frame_end.str() << "        return g_bus;\n    };\n";
    }
    body << events_code(all_stages, false);
    body << "    for (uint32_t base = 0; base < A.frames; base += OG_BUS_CHUNK) {\n"
         << "        const uint32_t n = min((uint32_t)OG_BUS_CHUNK, A.frames - base);\n"
         << (out.rings.empty() ? std::string() : "        cbase = base;\n" + cat(all_stages, &Codegen::Sect::chunk_begin))
         << "        if (n == OG_BUS_CHUNK && __all((int)(c.next_ev >= base + OG_BUS_CHUNK))) {\n"
         << "            // no lane of this wave has an event in the chunk: straight-line body\n";
    {
        const std::string mc = min_cnt(all_stages);
        std::string steady;
        for (int k : all_stages)
            for (const auto& fc : cg.sec[k].fast_conds) steady += (steady.empty() ? "" : " && ") + fc;
        // Sticky chunks in the ordinary kernel (OGC_STICKY1=0 turns them off): as in the pipelined kernels (emit_pipeline below), a wave stays in the quiet variant it is in while that
        // variant's conditions hold on the next chunk, instead of going back through the chunk loop's head, where the
        // compiler reconciles the register assignments of the four chunk bodies (fm_voice: 57 v_mov per 16-frame chunk).
        const bool sticky1 = !(ogabi::experiment_knob("OGC_STICKY1") && atoi(ogabi::experiment_knob("OGC_STICKY1")) == 0); // round 5: on (+4.5 % at 262 144 voices, +6 % at 1 M, +3.4 % saturator; profiles/r05a_session1.md)
        std::string stay_path1;
        auto variants = [&](const std::string& tail, const std::string& ind0) {
            auto quiet = [&](const std::string& flag, const std::string& ind1, const std::string& stay = std::string()) {
                const bool loop = sticky1 && !stay.empty();
                const std::string ind = loop ? ind1 + "    " : ind1;
                if (loop) body << ind1 << "for (;;) { // sticky: this variant again while its conditions hold\n";
                if (flag != "false, false") body << fc_sync(all_stages, ind);
                body << ind << "#pragma unroll " << unroll << "\n"
                     << ind << "for (uint32_t j = 0; j < OG_BUS_CHUNK; ++j) og::bus_put<TAPS" << (cg.bus_all_lanes ? ", !TAPS" : "")
                     << ">(A, c, bus, base + j, j, tick(base + j, og::BoolC<" << flag << tail << ">{}));\n";
                if (loop) {
                    body << ind << "const uint32_t base1 = base + OG_BUS_CHUNK;\n"
                         << ind << "if (!(base1 + OG_BUS_CHUNK <= A.frames && " << stay_path1 << (stay == "1" ? "" : " && " + stay) << ")) break;\n"
                         << ind << "og::bus_chunk_reduce(A, c, bus, base, OG_BUS_CHUNK);\n"
                         << ind << "base = base1;\n"
                         << (out.rings.empty() ? std::string() : ind + "cbase = base;\n" + cat(all_stages, &Codegen::Sect::chunk_begin))
                         << ind1 << "}\n";
                }
            };
            if (mc.empty()) {
                quiet("true, true", ind0, "1");
            } else {
                const std::string no_end = "__all((int)(" + mc + " > (uint32_t)OG_BUS_CHUNK))", no_rel = "__all((int)(" + rs_sum(all_stages) + " == 0.0f))";
                body << ind0 << "if (" << no_end << ") { // no envelope stage ends in this chunk\n"
                     << ind0 << "    if (" << no_rel << ") { // ... and no lane is in Release\n";
                quiet("false, false", ind0 + "        ", no_end + " && " + no_rel);
                body << ind0 << "    } else {\n";
                quiet("false, true", ind0 + "        ", no_end + " && !" + no_rel);
                body << ind0 << "    }\n" << ind0 << "} else {\n";
                quiet("true, true", ind0 + "    ");
                body << ind0 << "}\n";
            }
        };
        const std::string stay_entry1 = "__all((int)(c.next_ev >= base1 + OG_BUS_CHUNK))";
        if (steady.empty()) {
            stay_path1 = stay_entry1;
            variants("", "            ");
        } else {
            const std::string all_steady = "__all((int)(!c.valid || (" + steady + ")))";
            body << "            constexpr uint32_t CHUNK = OG_BUS_CHUNK;\n"
                 << "            if (" << all_steady << ") { // node steady states hold for the whole chunk\n";
            stay_path1 = stay_entry1 + " && " + all_steady;
            variants(", false, true", "                ");
            body << "            } else {\n";
            stay_path1 = stay_entry1 + " && !" + all_steady;
            variants("", "                ");
            body << "            }\n";
        }
    }
    body << "        } else {\n"
         << fc_sync(all_stages, "            ")
         << "            for (uint32_t j = 0; j < n; ++j) {\n"
         << "                events(base + j);\n"
         << "                og::bus_put<TAPS" << (cg.bus_all_lanes ? ", !TAPS" : "") << ">(A, c, bus, base + j, j, tick(base + j, og::BoolC<true>{}));\n"
         << "            }\n"
         << "        }\n"
         << "        og::bus_chunk_reduce(A, c, bus, base, n);\n"
         << "    }\n";
    body << "    og::bus_flush(A, c, bus);\n"
         << cat(all_stages, &Codegen::Sect::pre_store) << "    if (c.valid) {\n"
         << cat(all_stages, &Codegen::Sect::store) << vin_store();
    body << "    }\n    og::voice_end(A, c);\n}\n";

    // ---- pipelined kernels: workgroup = K waves over the same 64 voices ----------------------------
    // All waves take the same n_chunks + K - 1 steps, one barrier per step; at step t the wave of group
    // s works on hand-off chunk t - s.  A value produced by group a and read by group b lives in an LDS
    // ring of b - a + 1 chunks.  Which wave of the workgroup takes which group rotates with the
    // workgroup index, so that the SIMDs of a CU do not each collect one kind of stage.
    auto emit_pipeline = [&](const std::vector<std::vector<int>>& groups, int tag) {
        const int K = (int)groups.size();
        body << "\n// " << K << "-wave pipeline over the same 64 voices (small banks: more waves per SIMD).\n";
        for (int gi = 0; gi < K; ++gi) {
            body << "//   wave " << gi << ":";
            for (int ni : order)
                if (group_of(groups, cg.stage_of[ni]) == gi) body << " " << g.nodes[ni].name;
            body << (gi == K - 1 ? " + the mix bus\n" : "\n");
        }
        // FD_T = 0: one workgroup barrier per hand-off (every wave one chunk behind its predecessor); FD_T >= 2: flag
        // hand-off (og_kernel_rt.hip.h, handoff_wait / handoff_publish) over rings of FD_T chunks per crossing value
        body << "template <bool RAMPS, bool TAPS, uint32_t XCH_T = 0, uint32_t FD_T = 0>\n"
             << "__device__ __forceinline__ void voice_block_p" << tag << "(const OgBlockArgs& A)\n{\n"
             << "    __shared__ og::" << (out.voice_channels > 1 ? "BusLdsN<" + std::to_string(out.voice_channels) + ">" : std::string("BusLds")) << " bus;\n";
        {
            // frames per hand-off: 8.  16 (OGC_XCH16, two-wave pipeline only) measured +2% on a 94-block run at
            // 65 536 voices but -1.5% on the 188-block default run, and its doubled LDS rings leave room for only
            // four workgroups per CU: 98 304 voices (six per CU) ran 29% slower.  4 is 11% slower.
            size_t slots = 0;
            for (const auto& xv : cg.xvals) {
                int far = group_of(groups, xv.from);
                for (int u : xv.users) far = std::max(far, group_of(groups, u));
                const int depth = far - group_of(groups, xv.from) + 1;
                if (depth > 1) slots += (size_t)depth;
            }
            const bool wide = K == 2 && slots * 16 * 64 * 4 <= 36 * 1024 && ogabi::experiment_knob("OGC_XCH16");
            int xch = wide ? 16 : 8;
            // OGC_XCH=4|16: experiment for the next round -- with the sticky chunk loops a hand-off costs ~9 VALU + ~17 SALU
            // and a barrier instead of ~45 + ~40, so the trade between hand-off overhead (longer chunks) and lock-step wait /
            // LDS footprint (shorter ones) has moved since the measurements above.  OG_BUS_CHUNK (16) must stay a multiple.
            if (const char* ex = ogabi::experiment_knob("OGC_XCH")) {
                const int want = atoi(ex);
                if ((want == 4 || want == 8 || want == 16) && slots * (size_t)want * 64 * 4 <= 60 * 1024) xch = want;
            }
            // Round 5: the four-wave kernel waits more than it issues (54 % of its wave cycles), and 16-frame hand-offs -- half
            // the barriers -- measured +2.4 % at the driver's command and +3.3 % on 94-block regions at 65 536 voices; but
            // their LDS rings (fm_voice: 36.9 KB per workgroup) fit only four workgroups per CU, so a bank of five or six per
            // CU would run in two rounds.  Both exist: XCH_T = 16 instantiates the wide form (og_k4w_*), which the engine
            // launches when every workgroup of the bank is resident at once (og_engine.cpp, depth model).
            body << "    constexpr uint32_t XCH = XCH_T ? XCH_T : " << xch << "u; // frames per hand-off between the waves\n";
            if (K == 4 && xch == 8 && slots * 16 * 64 * 4 + 4160 * (size_t)std::max(1u, out.voice_channels) <= 40 * 1024) out.wide4 = true;
        }
        for (size_t k = 0; k < cg.xvals.size(); ++k) {
            const auto& xv = cg.xvals[k];
            int far = group_of(groups, xv.from);
            for (int u : xv.users) far = std::max(far, group_of(groups, u));
            const int depth = far - group_of(groups, xv.from) + 1;
            body << "    constexpr uint32_t XD" << k << " = " << (depth > 1 ? "FD_T ? FD_T : " : "") << depth << ";\n";
            if (depth > 1) body << "    __shared__ float chan" << k << "[XD" << k << "][XCH][OG_WAVE];\n";
        }
        body << "    __shared__ uint32_t prog[" << K << "]; // flag hand-off: chunks completed, per stage\n"
             << "    __shared__ uint32_t cons[" << K << "]; // ... and chunks whose inputs this stage has taken out of the rings\n"
             << "    if constexpr (FD_T != 0) {\n"
             << "        if (threadIdx.x < " << K << "u) prog[threadIdx.x] = cons[threadIdx.x] = 0u;\n"
             << "        __syncthreads();\n"
             << "    }\n";
        const char* rot_expr[5] = {"0u", "blockIdx.x", "(blockIdx.x >> 3)", "(blockIdx.x >> 5)", "(blockIdx.x >> 8)"};
        int rot = 1;
        if (const char* er = ogabi::experiment_knob("OGC_ROT")) rot = std::max(0, std::min(4, atoi(er)));
        body << "    const uint32_t stage = (threadIdx.x / OG_WAVE + " << rot_expr[rot] << ") % " << K << "u;\n"
             << "    og::VoiceCtx c;\n"
             << "    og::voice_begin_split<TAPS>(A, c);\n"
             << cg.common_decl.str() << "    if (c.valid) {\n" << cg.common_load.str() << "    }\n"
             << "    const uint32_t n_chunks = (A.frames + XCH - 1) / XCH;\n";
        for (int gi = 0; gi < K; ++gi) {
            const std::vector<int>& st = groups[gi];
            const bool last = gi == K - 1;
            int base_prio = 0;
            body << (gi == 0 ? "    if (stage == 0) {\n" : "    } else if (stage == " + std::to_string(gi) + ") {\n");
            {
                // VALU issue on a SIMD is arbitrated by priority, then age (MI355X_MICROARCH.md).  The first wave of
                // the pipeline (the light producer: envelopes, first operator) runs at priority 1: it never becomes
                // the straggler its consumers wait for at the hand-off barrier.  Measured, fm_voice: two waves at
                // 65 536 voices 0.0741 -> 0.0660 ms; four waves at 32 768 voices 0.0500 -> 0.0484 ms; raising the
                // other waves as well (2,1 / 3,2,1,0) is no better.  OGC_PRIO="p0,p1[,p2,p3]" overrides.
                // Round 4, four waves (depth-first order): the LAST wave (filter, bus) lowest, the first highest, the middle
                // ones between -- 2,1,1,0.  Interleaved A/B at the driver's command, 65 536 voices: 1,0,0,0 3.08e11; 0,0,0,0
                // 3.00e11; 0,0,1,0 3.20e11; 1,1,1,0 3.16e11; 2,1,1,0 3.26e11; 3,2,1,0 3.27e11; 3,2,2,0 3.28e11; 2,1,2,0 3.28e11;
                // 3,2,1,1 3.14e11 -- whatever is upstream must not wait for the wave that closes the chunk; the exact levels
                // above it are within noise.  (Neutral on the 188-block run, +1 % at 32 768 and 131 072 voices.)
                std::vector<int> pr = K >= 3 ? std::vector<int>{2, 1, 1, 0} : std::vector<int>{1, 0, 0, 0};
                if (K >= 3) {
                    pr.assign((size_t)K, 1);
                    pr.front() = 2;
                    pr.back() = 0;
                }
                if (const char* ep = ogabi::experiment_knob("OGC_PRIO")) {
                    pr.clear();
                    for (const char* q = ep; *q; ++q)
                        if (isdigit((unsigned char)*q)) pr.push_back(*q - '0');
                }
                base_prio = gi < (int)pr.size() ? pr[gi] : 0;
                // Round 6, flag hand-off (FD_T != 0): the order turns round -- 0,1,2,3, downstream first.  Without the barrier a
                // consumer that has its inputs runs them down at once and the producers fill in behind it; the rings stay near
                // empty and no wave sits blocked on a full one.  Interleaved A/B, 65 536 voices, 16-frame chunks, rings of 2
                // (profiles/r06_handoff_ab.md): barrier 2,1,1,0 (round 5) 5.45e11 at the driver's command; flags 2,1,1,0
                // 5.56e11; flags 0,0,0,0 5.40e11; flags 0,1,1,2 5.65e11; 0,0,1,2 5.74e11; 0,1,2,2 5.57e11; 1,1,2,3 5.38e11;
                // 0,0,2,3 5.32e11; flags 0,1,2,3 5.93e11 -- and the barrier with 0,1,2,3 4.89e11: the order only works without
                // the lock step.  Moving-cutoff variant 3.95e11 -> 4.37e11; 188-block regions 6.42e11 -> 6.2e11 (-3 %).
                const int flag_prio = ogabi::experiment_knob("OGC_PRIO") ? base_prio : std::min(gi, 3);
                body << "    constexpr int BASE_PRIO = FD_T != 0 ? " << flag_prio << " : " << base_prio << ";\n"
                     << "    og::set_prio<BASE_PRIO>();\n";
            }
            body << cat(st, &Codegen::Sect::decl) << "    if (c.valid) {\n" << cat(st, &Codegen::Sect::load) << "    }\n";
            body << "    auto derive = [&]() __attribute__((always_inline)) {\n" << cat(st, &Codegen::Sect::derive) << "    };\n";
            body << cat(st, &Codegen::Sect::pre) << "    derive();\n";
            // hand-off values this wave reads: fetched for the whole chunk before its first tick, so that the LDS
            // latency is paid once per chunk and not in front of every frame's first dependent instruction
            std::vector<size_t> reads;
            for (size_t k = 0; k < cg.xvals.size(); ++k) {
                const auto& xv = cg.xvals[k];
                bool used_here = false;
                for (int u : xv.users) used_here = used_here || group_of(groups, u) == gi;
                if (used_here && group_of(groups, xv.from) != gi) reads.push_back(k);
            }
            for (size_t k : reads) body << "    float xp" << k << "[XCH];\n";
            // Ramp-table rows (ramped inputs: the RAMPS variants; stream inputs: every variant).  `RV(row, slot)` / `ST(row)`
            // read A.ramp_table[row * stride + f] -- a scalar load behind a 64-bit address computation, per input and frame:
            // the launches that read the table took 1.8x the time of the others (SALU 2.9x per block, round 5 counters).  In
            // the unrolled chunk bodies the rows this wave reads are fetched for the WHOLE chunk at its top, next to the
            // hand-off values (consecutive words from a uniform base: wide scalar loads); the rolled event path keeps the
            // direct read.  RV -> RVP / ST -> STP in this wave's tick text.
            std::string tick_text = group_tick(groups, gi);
            std::set<int> rows;
            for (const char* mac : {"RV(", "ST("}) {
                const std::string pre_mac = std::string(mac, 2) + "P(";
                for (size_t pos = tick_text.find(mac); pos != std::string::npos; pos = tick_text.find(mac, pos + 1)) {
                    if (pos > 0 && (isalnum((unsigned char)tick_text[pos - 1]) || tick_text[pos - 1] == '_')) continue;
                    rows.insert(atoi(tick_text.c_str() + pos + 3));
                    tick_text.replace(pos, 3, pre_mac);
                }
            }
            for (int r : rows) body << "    float rv_" << r << "[XCH];\n";
            body << "    auto tick = [&](const uint32_t f, const uint32_t ch, const uint32_t j, auto chk) __attribute__((always_inline))"
                 << (last ? (out.voice_channels > 1 ? " -> og::OutN<" + std::to_string(out.voice_channels) + ">" : std::string(" -> float")) : std::string()) << " {\n"
                 << tick_text;
            if (last) body << "        return " << bus_expr << ";\n";
            body << "    };\n";
            body << events_code(st);
            // SALU instructions cost issue slots like VALU ones: the quiet chunk is a straight-line,
            // fully unrolled body; per-frame tests only exist on the (rare) event path
            auto call = [&](const std::string& flag) {
                const std::string t = std::string("tick(f, ch, j, og::BoolC<") + flag + ">{})";
                // row of the bus tile: f % OG_BUS_CHUNK = (base % OG_BUS_CHUNK) + j -- `base` is a multiple of XCH, which divides
                // OG_BUS_CHUNK, so there is no carry; written this way the chunk-invariant part is formed once per chunk and the
                // eight stores of the unrolled body take immediate offsets (one v_add + two SALU per frame less in the bus wave)
                return last ? "og::bus_put<TAPS>(A, c, bus, f, (base & (OG_BUS_CHUNK - 1u)) + j, " + t + ");" : t + ";";
            };
            // node steady-state conditions of this wave's stages (Sect::fast_conds), as in the ordinary kernel
            std::string steady;
            for (int k : st)
                for (const auto& fc : cg.sec[k].fast_conds) steady += (steady.empty() ? "" : " && ") + fc;
            // flags: stage-end checks, release arithmetic, hand-off values prefetched, node steady states
            // `stay` (sticky chunks): the condition -- on the NEXT chunk, `base1` -- under which this very variant runs again.
            // The wave then stays in a loop of its own around the variant's body (hand-off barrier inside) instead of
            // going back through the chunk loop's head.  Why: the chunk loop merges four bodies (two quiet variants, the
            // checked one, the rolled one) and the compiler brings every loop-carried value (envelope, phases, event
            // cursor: ~20 registers) back to one place after each of them -- 30 v_mov + ~40 SALU per wave and chunk, a sixth
            // of a quiet chunk's instructions (scripts/isa_blocks.py).  A loop with ONE body keeps its values where they are.
            // Static count, fm_voice four-wave kernel, release-free quiet chunk of waves 0-2: 231 + 230 + 222 -> ~201 + 187 + 194
            // VALU per 8 frames; rocprofv3 SQ_INSTS_VALU per 64-voice frame at the driver's command 120.7 -> 107.4.  Interleaved A/B on one MI355X, 65 536 voices: 94-block
            // runs 3.73e11 -> 4.01e11, the driver's command 3.29e11 -> 3.54e11, 131 072 voices 3.90e11 -> 4.20e11 (+7.5 % each);
            // parity subset of the GPU suite green with the variant before it became the default.  OGC_STICKY=0 turns it off.
            const bool sticky = !(ogabi::experiment_knob("OGC_STICKY") && atoi(ogabi::experiment_knob("OGC_STICKY")) == 0) && !ogabi::experiment_knob("OGC_FORCE_PATH");
            std::string stay_path; // conditions of the enclosing branches, on the next chunk
            // hand-off partners of this wave: the stages whose values it reads (producers) and the stages that read its values
            std::set<int> producers, consumers;
            for (const auto& xv : cg.xvals) {
                const int from = group_of(groups, xv.from);
                for (int u : xv.users) {
                    const int to = group_of(groups, u);
                    if (to == from) continue;
                    if (to == gi) producers.insert(from);
                    if (from == gi) consumers.insert(to);
                }
            }
            for (int q : producers) body << "    uint32_t seen_p" << q << " = 0u;\n";
            for (int q : consumers) body << "    uint32_t seen_c" << q << " = 0u;\n";
            // flags: before chunk `ch` -- its inputs are complete, and the ring slots it writes have been read
            std::string wait_all = "if constexpr (FD_T != 0) {";
            for (int q : producers) wait_all += " og::handoff_wait(&prog[" + std::to_string(q) + "], seen_p" + std::to_string(q) + ", ch + 1u);";
            for (int q : consumers) wait_all += " og::handoff_wait(&cons[" + std::to_string(q) + "], seen_c" + std::to_string(q) + ", ch + 1u - FD_T);";
            wait_all += " }";
            // (a stage that reads hand-off values gives their ring slots back as soon as the chunk's values sit in its registers --
            //  `taken`, after the prefetch of the unrolled bodies -- or, on the rolled path, with the chunk)
            const std::string taken = producers.empty() ? std::string()
                                                        : "if constexpr (FD_T != 0) og::handoff_publish(&cons[" + std::to_string(gi) + "], ch + 1u);";
            const std::string sync_line = "if constexpr (FD_T != 0) { og::handoff_publish(&prog[" + std::to_string(gi) + "], ch + 1u);" +
                                          (producers.empty() ? std::string() : " __hip_atomic_store(&cons[" + std::to_string(gi) + "], ch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);") +
                                          " } else OG_HANDOFF_BARRIER(); // hand-off: chunk `ch` of this stage is complete";
            const std::string bus_tail_fmt = // %B = first frame of the finished chunk, %N = its length
                "{ // the bus tile holds OG_BUS_CHUNK frames = OG_BUS_CHUNK / XCH hand-offs\n"
                "%I    const uint32_t lastf = %B + %N - 1;\n"
                "%I    if ((lastf % OG_BUS_CHUNK) == OG_BUS_CHUNK - 1 || lastf + 1 == A.frames)\n"
                "%I        og::bus_chunk_reduce(A, c, bus, lastf - (lastf % OG_BUS_CHUNK), (lastf % OG_BUS_CHUNK) + 1);\n"
                "%I}\n";
            auto bus_tail = [&](const std::string& ind, const std::string& n_expr) {
                std::string o = ind + bus_tail_fmt;
                auto sub = [&](const std::string& k, const std::string& v) {
                    for (size_t q = o.find(k); q != std::string::npos; q = o.find(k, q + v.size())) o.replace(q, k.size(), v);
                };
                sub("%I", ind);
                sub("%B", "base");
                sub("%N", n_expr);
                return o;
            };
            auto row_fetch = [&](const std::string& ind) { // the table rows of this chunk (`base` = its first frame within the launch)
                std::ostringstream o;
                if (rows.empty()) return std::string();
                bool only_ramps = true; // (ST rows -- stream inputs -- are read in every variant, RV rows only under RAMPS)
                for (size_t pos = tick_text.find("STP("); pos != std::string::npos; pos = std::string::npos) only_ramps = false;
                o << ind << (only_ramps ? "if (RAMPS) {\n" : "{\n");
                for (int r : rows)
                    o << ind << "    og::row_fetch<XCH>(rv_" << r << ", A.ramp_table + (size_t)" << r << " * A.ramp_stride + base);\n";
                o << ind << "}\n";
                return o.str();
            };
            auto quiet = [&](const char* chk_flag, const char* rel_flag, bool st_flag, const std::string& ind0, const std::string& stay = std::string()) {
                const bool pre = !reads.empty() || !rows.empty();
                const bool loop = sticky && !stay.empty();
                const std::string ind = loop ? ind0 + "    " : ind0;
                if (loop) body << ind0 << "for (;;) { // sticky: this variant again while its conditions hold\n";
                if (std::string(rel_flag) == "true") body << fc_sync(st, ind);
                if (pre) {
                    body << row_fetch(ind);
                    body << ind << "#pragma unroll\n" << ind << "for (uint32_t j = 0; j < XCH; ++j) {\n";
                    for (size_t k : reads) body << ind << "    xp" << k << "[j] = chan" << k << "[ch % XD" << k << "][j][c.lane];\n";
                    body << ind << "}\n";
                    if (!reads.empty() && !taken.empty()) body << ind << taken << "\n";
                }
                const std::string flag = std::string(chk_flag) + ", " + rel_flag + ", " + (pre ? "true" : "false") + ", " +
                                         (st_flag ? "true" : "false");
                body << ind << "#pragma unroll\n"
                     << ind << "for (uint32_t j = 0; j < XCH; ++j) {\n"
                     << ind << "    const uint32_t f = base + j;\n"
                     << ind << "    " << call(flag) << "\n"
                     << ind << "}\n";
                if (loop) {
                    body << ind << "const uint32_t ch1 = ch + 1u, base1 = base + XCH;\n"
                         << ind << "if (!(ch1 < n_chunks && A.frames - base1 >= (uint32_t)XCH && " << stay_path << (stay == "1" ? "" : " && " + stay)
                         << ")) break;\n";
                    if (last) body << bus_tail(ind, "XCH");
                    body << ind << sync_line << "\n"
                         << ind << "++t;\n" << ind << "ch = ch1;\n" << ind << "base = base1;\n";
                    if (!wait_all.empty()) body << ind << wait_all << "\n";
                    body << ind0 << "}\n";
                }
            };
            const std::string mc = min_cnt(st);
            // ---- events in the pipelined kernels -----------------------------------------------------------------
            // (1) A wave only has to leave the straight-line path for events IT handles.  Every wave of the pipeline
            //     walks the voice's event list, but e.g. FMVoice's second wave (operators 2 and 1, filter) has no
            //     handler for `gate` -- only the envelope wave has -- and used to run the chunk frame by frame for
            //     nothing.  Events without a handler in this wave's stages are consumed at the top of the chunk.
            // (2) A chunk that does hold an event (or an envelope stage end) runs the UNROLLED body with one
            //     wave-uniform `any lane has an event on this frame` test per frame, instead of a rolled frame loop
            //     that serialises the frames (measured: the rolled event chunk costs twice a quiet one, and the launch
            //     ends with the wave that met the most events).  It replaces the stage-end-check variant of the quiet
            //     chunk, so the kernel does not grow.
            // Measured (scripts/dbg_event_cost.py, fm_voice, 65 536 voices, kernel ms per 256-frame block, round-2 form ->
            // both): one retrigger per voice in 5 120 frames 0.0641 -> 0.0623, two 0.0714 -> 0.0677 (the cost of an
            // event falls by a third, nearly all of it from (1)) -- but (2) inlines the handlers into every frame of the
            // checked body (two-wave kernel 26.8 -> 35.9 KB) and the QUIET path pays for it: idle bank 0.0514 -> 0.0528,
            // sustaining 0.0536 -> 0.0553.  At the benchmark's event density the two cancel (2.73e11 either way), so
            // (1) was on and (2) off through round 3: OGC_EVSKIP=0 turns (1) off.
            const bool ev_skip = !(ogabi::experiment_knob("OGC_EVSKIP") && atoi(ogabi::experiment_knob("OGC_EVSKIP")) == 0);
            // (round 4: (2) is ON -- with the event records prefetched into registers (og::VoiceCtx::nx_*) the handlers inlined
            //  into the checked body no longer carry loads and waits; four-wave kernel at 65 536 voices, interleaved A/B:
            //  3.07e11 either way at the driver's command, 3.57e11 against 3.44e11 on the 188-block run; OGC_EVUNROLL=0 turns it off)
            const bool ev_unroll = !(ogabi::experiment_knob("OGC_EVUNROLL") && atoi(ogabi::experiment_knob("OGC_EVUNROLL")) == 0);
            std::string relevant; // condition on `tgt`: this wave has a handler for the event
            {
                const std::string code = cat(st, &Codegen::Sect::derive) + group_tick(groups, gi) + cat(st, &Codegen::Sect::decl);
                for (size_t i = 0; i < out.inputs.size(); ++i) {
                    const InputInfo& in = out.inputs[i];
                    if (!(in.decl.kind == Kind::Value && in.decl.per_voice)) continue;
                    const std::string var = "vin_" + std::to_string(i);
                    bool used = false;
                    for (size_t p = code.find(var); p != std::string::npos && !used; p = code.find(var, p + 1)) {
                        const size_t e = p + var.size();
                        used = e >= code.size() || !isdigit((unsigned char)code[e]);
                    }
                    // (the first wave stores the per-voice inputs back: it applies every SETVALUE)
                    if (used || gi == 0) relevant += (relevant.empty() ? "" : " || ") + ("tgt == (OG_EV_SETVALUE | " + std::to_string(i) + "u)");
                }
                std::set<int> handled;
                for (int k : st)
                    for (auto& kv : cg.sec[k].ev_handlers)
                        if (!kv.second.str().empty()) handled.insert(kv.first);
                for (int h : handled) relevant += (relevant.empty() ? "" : " || ") + ("tgt == " + std::to_string(h) + "u");
                if (relevant.empty()) relevant = "false";
            }
            const char* force = ogabi::experiment_knob("OGC_FORCE_PATH"); // experiment knob: b | c | ev -- quiet chunks take the release-arithmetic / stage-end-check / event path (results stay valid)
            // the checked chunk: unrolled, stage-end checks and release arithmetic on, events applied on their frame
            auto checked = [&](const std::string& ind) {
                const bool pre = !reads.empty() || !rows.empty();
                body << fc_sync(st, ind);
                if (pre) {
                    body << row_fetch(ind);
                    body << ind << "#pragma unroll\n" << ind << "for (uint32_t j = 0; j < XCH; ++j) {\n";
                    for (size_t k : reads) body << ind << "    xp" << k << "[j] = chan" << k << "[ch % XD" << k << "][j][c.lane];\n";
                    body << ind << "}\n";
                    if (!reads.empty() && !taken.empty()) body << ind << taken << "\n";
                }
                const std::string flag = std::string("true, true, ") + (pre ? "true" : "false") + ", false";
                body << ind << "#pragma unroll\n"
                     << ind << "for (uint32_t j = 0; j < XCH; ++j) {\n"
                     << ind << "    const uint32_t f = base + j;\n"
                     << ind << "    if (__any((int)(f == c.next_ev))) events(f); // (wave-uniform: rare)\n"
                     << ind << "    " << call(flag) << "\n"
                     << ind << "}\n";
            };
            auto variants = [&](bool st_flag, const std::string& ind0) {
                if (force && force[0] == 'c') {
                    quiet("true", "true", st_flag, ind0);
                } else if (mc.empty()) {
                    quiet("true", "true", st_flag, ind0, "1");
                } else {
                    const std::string no_end = "__all((int)(" + mc + " > (uint32_t)XCH))", no_rel = "__all((int)(" + rs_sum(st) + " == 0.0f))";
                    body << ind0 << "if (" << no_end << ") { // no envelope stage ends in this chunk\n"
                         << ind0 << "    if (" << no_rel << ") { // ... and no lane is in Release\n";
                    quiet("false", force && force[0] == 'b' ? "true" : "false", st_flag, ind0 + "        ", no_end + " && " + no_rel);
                    body << ind0 << "    } else {\n";
                    quiet("false", "true", st_flag, ind0 + "        ", no_end + " && !" + no_rel);
                    body << ind0 << "    }\n" << ind0 << "} else {\n";
                    quiet("true", "true", st_flag, ind0 + "    ");
                    body << ind0 << "}\n";
                }
            };
            // the two straight-line variants only (the caller has established that no countdown ends in the chunk)
            const int rel_prio = ogabi::experiment_knob("OGC_RELPRIO") ? atoi(ogabi::experiment_knob("OGC_RELPRIO")) : -1;
            auto fast_variants = [&](bool st_flag, const std::string& ind0) {
                const std::string no_rel = "__all((int)(" + rs_sum(st) + " == 0.0f))";
                body << ind0 << "if (" << no_rel << ") { // no lane is in Release\n";
                quiet("false", force && force[0] == 'b' ? "true" : "false", st_flag, ind0 + "    ", no_rel);
                body << ind0 << "} else {\n";
                // OGC_RELPRIO=n (experiment for grouped banks, og_group_voices: the waves that release are then few and
                // whole, and their workgroups are the ones a launch waits for; measured WITHOUT grouping in round 4: loses)
                if (rel_prio >= 0) body << ind0 << "    __builtin_amdgcn_s_setprio(" << rel_prio << ");\n";
                quiet("false", "true", st_flag, ind0 + "    ", "!" + no_rel);
                if (rel_prio >= 0) body << ind0 << "    og::set_prio<BASE_PRIO>();\n";
                body << ind0 << "}\n";
            };
            if (last) body << "    og::bus_init(c, bus); // (this wave owns the tile)\n";
            body << "    for (uint32_t t = 0; t < n_chunks + (FD_T ? 0u : " << (K - 1) << "u); ++t) {\n"
                 << "        " << (sticky ? "" : "const ") << "uint32_t ch = t - (FD_T ? 0u : " << gi << "u);\n"
                 << "        if (ch < n_chunks) {\n";
            if (!wait_all.empty()) body << "        " << wait_all << "\n";
            body
                 << "        " << (sticky ? "" : "const ") << "uint32_t base = ch * XCH;\n"
                 << "        const uint32_t n = min((uint32_t)XCH, A.frames - base);\n";
            if (ev_skip && relevant != "true")
                body << "        if (!__all((int)(c.next_ev >= base + XCH))) { // consume the events this wave has no handler for\n"
                     << "            while (c.next_ev < base + XCH) {\n"
                     << "                const uint32_t tgt = c.nx_target;\n"
                     << "                if (" << relevant << ") break;\n"
                     << "                og::ev_advance(A, c);\n"
                     << "            }\n"
                     << "        }\n";
            // A wave that meets an event or an envelope stage end runs the checked body for that chunk -- about twice the
            // work of a quiet one -- and the other waves of its workgroup wait for it at the hand-off barrier: it runs
            // that chunk at the highest priority and drops back afterwards.  Interleaved A/B, fm_voice, 65 536 voices,
            // 94-block runs, nine pairs: +0.9 .. +5.8 %, mean +2.1 % (3.68e11 -> 3.76e11); level 2 instead of 3 gives a
            // third of it; neutral at 32 768 and 262 144 voices; raising the release-arithmetic variant as well loses
            // the gain.  (The two-wave kernel carries it too although it does nothing for it -- forced with
            // OSCEN_GPU_SPLIT=2 at 65 536 voices: 3.16e11 with, 3.19e11 without; the engine does not pick that kernel at
            // any bank size of the bench.)  OGC_SLOWPRIO=-1 turns it off.
            const int slow_prio = ogabi::experiment_knob("OGC_SLOWPRIO") ? atoi(ogabi::experiment_knob("OGC_SLOWPRIO")) : 3;
            const bool one_checked = ev_unroll && !mc.empty() && !force; // events and stage ends share ONE unrolled, checked body
            if (one_checked)
                body << "        if (n == XCH && __all((int)(c.next_ev >= base + XCH)) && __all((int)(" << mc
                     << " > (uint32_t)XCH))) { // nothing happens in this chunk: no event, no envelope stage end\n";
            else
                body << "        if (n == XCH && __all((int)(c.next_ev >= base + XCH))" << (force && force[0] == 'e' ? " && A.frames == 0u" : "") << ") {\n";
            auto pick = [&](bool st_flag, const std::string& ind) {
                if (one_checked) fast_variants(st_flag, ind);
                else variants(st_flag, ind);
            };
            const std::string stay_entry = std::string("__all((int)(c.next_ev >= base1 + XCH))") +
                                           (one_checked ? " && __all((int)(" + mc + " > (uint32_t)XCH))" : std::string());
            if (steady.empty()) {
                stay_path = stay_entry;
                pick(false, "            ");
            } else {
                const std::string all_steady = "__all((int)(!c.valid || (" + steady + ")))";
                body << "            constexpr uint32_t CHUNK = XCH;\n"
                     << "            if (" << all_steady << ") { // node steady states hold for the whole chunk\n";
                stay_path = stay_entry + " && " + all_steady;
                pick(true, "                ");
                body << "            } else {\n";
                stay_path = stay_entry + " && !" + all_steady;
                pick(false, "                ");
                body << "            }\n";
            }
            if (one_checked && ogabi::experiment_knob("OGC_STAGEEND_BODY") && atoi(ogabi::experiment_knob("OGC_STAGEEND_BODY")) != 0) {
                // experiment (round 6): a chunk with a stage end but NO event takes a body with the stage-end checks and without
                // the per-frame event tests and the inlined handlers
                body << "        } else if (n == XCH && __all((int)(c.next_ev >= base + XCH))) { // a stage end, no event\n";
                if (slow_prio >= 0) body << "            __builtin_amdgcn_s_setprio(" << slow_prio << ");\n";
                quiet("true", "true", false, "            ");
                if (slow_prio >= 0) body << "            og::set_prio<BASE_PRIO>();\n";
            }
            if (one_checked) {
                body << "        } else if (n == XCH) { // an event or a stage end in this chunk: the checked, unrolled body\n";
                if (slow_prio >= 0) body << "            __builtin_amdgcn_s_setprio(" << slow_prio << "); // the wave on the slow path is the straggler\n";
                checked("            ");
                if (slow_prio >= 0) body << "            og::set_prio<BASE_PRIO>();\n";
            }
            body << "        } else {\n"
                 << fc_sync(st, "            ")
                 << "            for (uint32_t j = 0; j < n; ++j) {\n"
                 << "                const uint32_t f = base + j;\n"
                 << "                events(f);\n"
                 << "                " << call("true") << "\n"
                 << "            }\n"
                 << "        }\n";
            if (last) body << bus_tail("        ", "n");
            body << "        }\n"
                 << "        " << sync_line << "\n"
                 << "    }\n";
            if (last) body << "    og::bus_flush(A, c, bus);\n";
            body << cat(st, &Codegen::Sect::pre_store) << "    if (c.valid) {\n" << cat(st, &Codegen::Sect::store)
                 << (gi == 0 ? vin_store() : std::string()) << "    }\n";
            if (gi == 0) body << "    og::voice_end(A, c);\n";
        }
        body << "    }\n}\n";
    };
    if (!cg.groups2.empty()) emit_pipeline(cg.groups2, 2);
    if (!cg.groups4.empty()) emit_pipeline(cg.groups4, 4);

    std::string user_src;
    for (const auto& kv : cg.user_fns) user_src += kv.second;
    const std::string body_s = user_src + body.str();
    // the wide four-wave form: 16-frame chunks, flag hand-off over rings of two chunks (round 6; round 5: one barrier per chunk).
    // OGC_FLAGS="xch,depth" (experiment): `xch` frames per chunk, rings of `depth` chunks, depth 0 = the barrier
    std::string wide_args = "16, 2";
    if (const char* ef = ogabi::experiment_knob("OGC_FLAGS")) {
        int xch = 0, fd = 0;
        if (sscanf(ef, "%d,%d", &xch, &fd) == 2 && (xch == 4 || xch == 8 || xch == 16) && (fd == 0 || (fd >= 2 && fd <= 8)))
            wide_args = std::to_string(xch) + ", " + std::to_string(fd);
    }
    out.hash = fnv1a(body_s + "|lpv" + std::to_string(out.lpv) + (cg.ev_capacity != 2 ? "|evq" + std::to_string(cg.ev_capacity) : std::string()) +
                     "|wide" + wide_args + "|rt" + OG_RT_DIGEST);
    char hs[32];
    snprintf(hs, sizeof hs, "%016llx", (unsigned long long)out.hash);

    std::ostringstream src;
    src << "// GENERATED by oscen_amd/csrc/og_graph.cpp from graph '" << g.name << "' -- do not edit.\n"
        << "// One fused voice kernel: " << out.state.size() << " state words/voice, " << out.n_slots
        << " uniform slots, " << out.n_ramps << " ramped inputs, " << out.n_event_inputs << " event inputs"
        << (out.lpv > 1 ? ", " + std::to_string(out.lane_state.size()) + " arrays x " + std::to_string(out.lpv) + " lanes x " +
                              std::to_string(out.lane_width) + " words/voice"
                        : std::string())
        << ".\n"
        << "// Node order: ";
    if (out.node_order.empty()) out.node_order = out.schedule_order; // (not reordered: emitted in the reference's order)
    for (auto& nn : out.node_order) src << nn << " ";
    if (out.schedule_order != out.node_order) {
        src << "\n// Schedule (depth first from the sinks; same values, any topological order of a pure dataflow graph): ";
        for (auto& nn : out.schedule_order) src << nn << " ";
    }
    if (out.lpv > 1) src << "\n#define OG_HPL " << out.lane_width << " // harmonics per lane (OGC_HPL)";
    if (cg.ev_capacity != 2) src << "\n#define OG_NODE_EVENTS_PER_FRAME " << cg.ev_capacity << " // event_queue_capacity of a node type of this graph";
    src << "\n#include \"og_kernel_rt.hip.h\"\n#include \"og_nodes.hip.h\"\n\n"
        << "#if OG_NODE_EVENTS_PER_FRAME <= 4\n#define OG_EV_LOOP_PRAGMA _Pragma(\"unroll\")\n#else\n#define OG_EV_LOOP_PRAGMA _Pragma(\"unroll 1\")\n#endif\n"
        << "#define SF(i) og::slot_f(A, (i))\n#define SU(i) og::slot_u(A, (i))\n"
        << "#define RV(row, slot) (RAMPS ? A.ramp_table[(size_t)(row) * A.ramp_stride + f] : og::slot_f(A, (slot)))\n"
        << "#define ST(row) A.ramp_table[(size_t)(row) * A.ramp_stride + f]\n"
        << "// the pipelined kernels: the chunk's rows are in rv_<row>[] when the chunk body is the unrolled one (BoolC::pre)\n"
        << "#define RVP(row, slot) (RAMPS ? og::row_pick<decltype(chk)::pre>(rv_##row, j, A, (row), f) : og::slot_f(A, (slot)))\n"
        << "#define STP(row) og::row_pick<decltype(chk)::pre>(rv_##row, j, A, (row), f)\n\n"
        << "namespace og_gen_" << hs << " {\n"
        << "constexpr int LPV = " << out.lpv << "; // lanes per voice\n"
        << body_s << "} // namespace\n\n#undef SF\n#undef SU\n#undef RV\n#undef ST\n#undef RVP\n#undef STP\n\n";
    const char* variants[4][3] = {{"00", "false", "false"}, {"10", "true", "false"}, {"01", "false", "true"},
                                  {"11", "true", "true"}};
    // register budget of the ordinary kernel: 4 waves per SIMD = 128 VGPRs.  The 4-lanes-per-voice e-piano form
    // (OG_HPL = 8) keeps eight harmonics of nine arrays per lane.  Round 3 gave it the two-waves-per-SIMD budget (201
    // VGPRs, no spill; the three-wave budget measured +1.7 % with 80 spills and was not taken).  Round 5: decay / release /
    // mult are no longer register state (og_nodes.hip.h, EpAmp) -- see scripts/isa_mix.py epiano_voice.  Round 4: with the
    // packed-fma bodies the three-wave budget (168 VGPRs) spills 76 registers, all of them around the chunk loop (25
    // scratch instructions per 16-frame chunk, none in the frame loop: scripts/isa_mix.py epiano_voice OGC_WAVES_EU=3)
    // and is 3.3 % faster in an interleaved A/B (1.57e11 -> 1.62e11 at 262 144 voices): taken.  Four waves (128 VGPRs)
    // spill inside the frame loop.
    int waves_eu = (out.lpv > 1 && out.lane_width == 8) ? 3 : 4;
    if (const char* ew = ogabi::experiment_knob("OGC_WAVES_EU")) waves_eu = std::max(1, std::min(8, atoi(ew)));
    for (auto& v : variants)
        src << "extern \"C\" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(" << waves_eu << "))) void og_k_" << hs << "_" << v[0]
            << "(OgBlockArgs A) { og_gen_" << hs << "::voice_block<" << v[1] << ", " << v[2] << ">(A); }\n";
    std::vector<std::pair<int, int>> depths; // (tag: what OgBlockArgs::split selects, waves per workgroup)
    if (!cg.groups2.empty()) depths.push_back({2, (int)cg.groups2.size()});
    if (!cg.groups4.empty()) depths.push_back({4, (int)cg.groups4.size()});
    // OGC_NARROW_FD=<depth> (experiment): the 8-frame shapes with the flag hand-off as well, rings of `depth` chunks
    std::string narrow_args;
    if (const char* en = ogabi::experiment_knob("OGC_NARROW_FD")) {
        const int fd = atoi(en);
        if (fd >= 2 && fd <= 8) narrow_args = ", 8, " + std::to_string(fd);
    }
    for (auto [K, W] : depths)
        for (auto& v : variants)
            src << "extern \"C\" __global__ __launch_bounds__(" << 64 * W << ") void og_k" << K << "_" << hs << "_" << v[0]
                << "(OgBlockArgs A) { og_gen_" << hs << "::voice_block_p" << K << "<" << v[1] << ", " << v[2] << narrow_args << ">(A); }\n";
    if (out.wide4)
        for (auto& v : variants)
            src << "extern \"C\" __global__ __launch_bounds__(" << 64 * (int)cg.groups4.size() << ") void og_k4w_" << hs << "_" << v[0]
                << "(OgBlockArgs A) { og_gen_" << hs << "::voice_block_p4<" << v[1] << ", " << v[2] << ", " << wide_args << ">(A); }\n";
    src << "\n#ifndef OG_JIT\n#include \"og_registry.h\"\n"
        << "static void og_launch_" << hs << "(const OgBlockArgs& A, bool ramps, bool taps, hipStream_t s)\n{\n"
        << "    const dim3 grid(((size_t)A.n_voices * " << out.lpv << " + A.lanes - 1) / A.lanes), block(OG_WAVE);\n";
    if (out.wide4)
        src << "    if (A.split == 4u && A.wide) { // four waves per 64 voices, 16-frame hand-offs\n"
            << "        const dim3 gk((A.n_voices + OG_WAVE - 1) / OG_WAVE), bk(" << cg.groups4.size() << " * OG_WAVE);\n"
            << "        if (!ramps && !taps) hipLaunchKernelGGL(og_k4w_" << hs << "_00, gk, bk, 0, s, A);\n"
            << "        else if (ramps && !taps) hipLaunchKernelGGL(og_k4w_" << hs << "_10, gk, bk, 0, s, A);\n"
            << "        else if (!ramps && taps) hipLaunchKernelGGL(og_k4w_" << hs << "_01, gk, bk, 0, s, A);\n"
            << "        else hipLaunchKernelGGL(og_k4w_" << hs << "_11, gk, bk, 0, s, A);\n"
            << "        return;\n    }\n";
    for (auto [K, W] : depths)
        src << "    if (A.split == " << K << "u) { // " << W << " waves per 64 voices\n"
            << "        const dim3 gk((A.n_voices + OG_WAVE - 1) / OG_WAVE), bk(" << W << " * OG_WAVE);\n"
            << "        if (!ramps && !taps) hipLaunchKernelGGL(og_k" << K << "_" << hs << "_00, gk, bk, 0, s, A);\n"
            << "        else if (ramps && !taps) hipLaunchKernelGGL(og_k" << K << "_" << hs << "_10, gk, bk, 0, s, A);\n"
            << "        else if (!ramps && taps) hipLaunchKernelGGL(og_k" << K << "_" << hs << "_01, gk, bk, 0, s, A);\n"
            << "        else hipLaunchKernelGGL(og_k" << K << "_" << hs << "_11, gk, bk, 0, s, A);\n"
            << "        return;\n    }\n";
    src << "    if (!ramps && !taps) hipLaunchKernelGGL(og_k_" << hs << "_00, grid, block, 0, s, A);\n"
        << "    else if (ramps && !taps) hipLaunchKernelGGL(og_k_" << hs << "_10, grid, block, 0, s, A);\n"
        << "    else if (!ramps && taps) hipLaunchKernelGGL(og_k_" << hs << "_01, grid, block, 0, s, A);\n"
        << "    else hipLaunchKernelGGL(og_k_" << hs << "_11, grid, block, 0, s, A);\n}\n"
        << "static int og_occ_" << hs << "(int depth) // resident workgroups per CU of each shape (registers, LDS)\n{\n    int n = 0;\n";
    for (auto [K, W] : depths)
        src << "    if (depth == " << K << ") { (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, og_k" << K << "_" << hs << "_00, " << 64 * W
            << ", 0); return n; }\n";
    if (out.wide4)
        src << "    if (depth == 5) { (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, og_k4w_" << hs << "_00, " << 64 * (int)cg.groups4.size()
            << ", 0); return n; } // (5 = the wide four-wave form)\n";
    src << "    if (depth <= 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, og_k_" << hs << "_00, OG_WAVE, 0);\n    return n;\n}\n"
        << "static const OgKernelRegistrar og_reg_" << hs << "(0x" << hs << "ull, \"" << g.name << "\", &og_launch_"
        << hs << ", &og_occ_" << hs << ");\n#endif\n";
    out.source = src.str();
    return cgp;
}

} // namespace ogc
