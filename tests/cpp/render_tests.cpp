// render_tests.cpp -- the reference's integration tests, written against the C++ host mirror.
//
// Same structure as the reference's tests/render_prim.rs, tests/ext_input.rs and tests/load_effect.rs:
// a MyClient that forwards rendered audio, test_setup() building Dispatch::new(renderer, client), graphs
// built by dispatching OscRouteGraph messages, one RenderRange, exact comparison with an array literal.
// Where the reference constructs `SparkleRenderer::default()`, this constructs the renderer plugin named
// by $FRIENDSHIP_RENDERER_LIB (the HIP engine on a GPU box; the CPU oracle in the CPU-only test run).
//
// Build: g++ -std=c++17 -O1 -o render_tests render_tests.cpp -ldl      Run: ./render_tests [test-name]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <deque>
#include <functional>
#include <iostream>
#include <memory>

#include "../../libfriendship_amd/host/friendship.hpp"

using namespace friendship;
using namespace friendship::routing;
using friendship::dispatch::OscRenderer;
using friendship::dispatch::OscResMan;
using friendship::dispatch::OscRouteGraph;

// `struct MyClient { tx: Sender<Array2<f32>> }`: the channel is a queue shared with the test body.
using Channel = std::shared_ptr<std::deque<Array2>>;
struct MyClient : Client {
    Channel tx;
    explicit MyClient(Channel c) : tx(std::move(c)) {}
    void audio_rendered(Array2 buffer, uint64_t) override { tx->push_back(std::move(buffer)); }
};

using TestDispatch = Dispatch<std::unique_ptr<render::Renderer>, MyClient>;

static std::pair<TestDispatch, Channel> test_setup() {
    const char *lib = std::getenv("FRIENDSHIP_RENDERER_LIB");
    if (!lib) throw std::runtime_error("set FRIENDSHIP_RENDERER_LIB to the renderer plugin (.so) under test");
    Channel rx = std::make_shared<std::deque<Array2>>();
    std::unique_ptr<render::Renderer> r = std::make_unique<render::PluginRenderer>(lib);
    return {TestDispatch(std::move(r), MyClient(rx)), rx};
}

static Array2 recv(const Channel &rx) {
    if (rx->empty()) throw std::runtime_error("nothing was rendered");
    Array2 a = std::move(rx->front());
    rx->pop_front();
    return a;
}

static Array2 array(std::initializer_list<float> row) { return Array2{1, row.size(), std::vector<float>(row)}; }

#define ASSERT_EQ_ARR(got, want)                                                                  \
    do {                                                                                          \
        Array2 g_ = (got), w_ = (want);                                                           \
        if (!(g_ == w_)) {                                                                        \
            std::cerr << __FILE__ << ":" << __LINE__ << ": assertion failed: got [";             \
            for (float v : g_.data) std::cerr << v << " ";                                        \
            std::cerr << "] expected [";                                                          \
            for (float v : w_.data) std::cerr << v << " ";                                        \
            std::cerr << "]\n";                                                                   \
            throw std::runtime_error("assert_eq failed");                                         \
        }                                                                                         \
    } while (0)

static EffectId prim_id(const char *name, const char *url) { return EffectId::make(name, std::nullopt, {url}); }
static EffectId delay_id() { return prim_id("Delay", "primitive:///Delay"); }
static EffectId sum2_id() { return prim_id("Sum2", "primitive:///Sum2"); }
static EffectId const_id() { return prim_id("F32Constant", "primitive:///F32Constant"); }
static EffectId mult_id() { return prim_id("Multiply", "primitive:///Multiply"); }
static EffectId div_id() { return prim_id("Divide", "primitive:///Divide"); }
static EffectId mod_id() { return prim_id("Modulo", "primitive:///Modulo"); }
static EffectId min_id() { return prim_id("Minimum", "primitive:///Minimum"); }

static OscRenderer::RenderRange render_range(uint64_t start, uint64_t end, uint32_t slots, Jagged2 inputs = {}) {
    return OscRenderer::RenderRange{start, end, slots, std::move(inputs)};
}

// ---- tests/render_prim.rs ------------------------------------------------------------------------
static void render_zeros() {
    auto [dispatch, rx] = test_setup();
    dispatch.dispatch(render_range(0, 4, 1));
    ASSERT_EQ_ARR(recv(rx), array({0.f, 0.f, 0.f, 0.f}));
}

static void render_const() {
    auto [dispatch, rx] = test_setup();
    auto handle = NodeHandle::make(1);
    dispatch.dispatch(OscRouteGraph::AddNode{handle, const_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(handle, EdgeWeight::make(f32_to_bits(0.5f), 0))});
    dispatch.dispatch(render_range(0, 4, 1));
    ASSERT_EQ_ARR(recv(rx), array({0.5f, 0.5f, 0.5f, 0.5f}));
}

static void render_delay() {
    auto [dispatch, rx] = test_setup();
    auto delay_hnd = NodeHandle::make(1);
    dispatch.dispatch(OscRouteGraph::AddNode{delay_hnd, delay_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(delay_hnd, EdgeWeight::make(0, 0))});
    auto const_hnd = NodeHandle::make(2);
    dispatch.dispatch(OscRouteGraph::AddNode{const_hnd, const_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(const_hnd, delay_hnd, EdgeWeight::make(f32_to_bits(0.5f), 0))});
    const_hnd = NodeHandle::make(3);
    dispatch.dispatch(OscRouteGraph::AddNode{const_hnd, const_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(const_hnd, delay_hnd, EdgeWeight::make(f32_to_bits(2.f), 1))});
    dispatch.dispatch(render_range(0, 4, 1));
    ASSERT_EQ_ARR(recv(rx), array({0.f, 0.f, 0.5f, 0.5f}));
}

static void binop(EffectId id, float a, float b, float exp) {
    auto [dispatch, rx] = test_setup();
    auto op_hnd = NodeHandle::make(1);
    dispatch.dispatch(OscRouteGraph::AddNode{op_hnd, id});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(op_hnd, EdgeWeight::make(0, 0))});
    auto const_hnd = NodeHandle::make(2);
    dispatch.dispatch(OscRouteGraph::AddNode{const_hnd, const_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(const_hnd, op_hnd, EdgeWeight::make(f32_to_bits(a), 0))});
    const_hnd = NodeHandle::make(3);
    dispatch.dispatch(OscRouteGraph::AddNode{const_hnd, const_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(const_hnd, op_hnd, EdgeWeight::make(f32_to_bits(b), 1))});
    dispatch.dispatch(render_range(0, 4, 1));
    ASSERT_EQ_ARR(recv(rx), array({exp, exp, exp, exp}));
}
static void render_mult() { binop(mult_id(), 0.5f, -3.f, -1.5f); }
static void render_sum2() { binop(sum2_id(), 0.5f, -3.f, -2.5f); }
static void render_div() { binop(div_id(), 0.5f, -3.f, 0.5f / -3.0f); }
static void render_mod() { binop(mod_id(), -3.5f, 2.f, 0.5f); }
static void render_min() { binop(min_id(), -3.5f, 2.f, -3.5f); }

// ---- tests/ext_input.rs --------------------------------------------------------------------------
static void ext_render_passthrough() {
    auto [dispatch, rx] = test_setup();
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(NodeHandle::toplevel(), EdgeWeight::make(0, 0))});
    Jagged2 builder;
    builder.extend({1.f, 2.f, 3.f, 4.f});
    dispatch.dispatch(render_range(0, 4, 1, builder));
    ASSERT_EQ_ARR(recv(rx), array({1.f, 2.f, 3.f, 4.f}));
    builder = Jagged2();
    builder.extend({0.f, 1.f, 2.f});
    dispatch.dispatch(render_range(4, 8, 1, builder));
    ASSERT_EQ_ARR(recv(rx), array({0.f, 1.f, 2.f, 2.f}));   // empty inputs take on their last known value
    dispatch.dispatch(render_range(0, 4, 1));
    ASSERT_EQ_ARR(recv(rx), array({0.f, 0.f, 0.f, 0.f}));   // seeking implicitly zeros the inputs
}

static void ext_render_delay() {
    auto [dispatch, rx] = test_setup();
    auto delay_hnd = NodeHandle::make(1);
    dispatch.dispatch(OscRouteGraph::AddNode{delay_hnd, delay_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(delay_hnd, EdgeWeight::make(0, 0))});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_from_null(delay_hnd, EdgeWeight::make(0, 0))});
    Jagged2 builder;
    builder.extend({1.f, 2.f, 3.f, 4.f});
    dispatch.dispatch(render_range(0, 4, 1, builder));
    ASSERT_EQ_ARR(recv(rx), array({1.f, 2.f, 3.f, 4.f}));
    auto const_hnd = NodeHandle::make(2);
    dispatch.dispatch(OscRouteGraph::AddNode{const_hnd, const_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(const_hnd, delay_hnd, EdgeWeight::make(f32_to_bits(1.f), 1))});
    builder = Jagged2();
    builder.extend({1.f, 2.f, 3.f, 4.f});
    dispatch.dispatch(render_range(4, 8, 1, builder));
    ASSERT_EQ_ARR(recv(rx), array({4.f, 1.f, 2.f, 3.f}));
}

// ---- tests/load_effect.rs ------------------------------------------------------------------------
static EffectDesc create_multby2() {
    auto mult_hnd = NodeHandle::make(1);
    auto mult_data = EffectId::make("Multiply", std::nullopt, {"primitive:///Multiply"});
    auto const_hnd = NodeHandle::make(2);
    auto const_data = EffectId::make("Constant", std::nullopt, {"primitive:///F32Constant"});
    AdjList list;
    list.nodes = {{mult_hnd, mult_data}, {const_hnd, const_data}};
    list.edges = {Edge::new_from_null(mult_hnd, EdgeWeight::make(0, 0)),                            // input -> multiply (A)
                  Edge::new_to_null(mult_hnd, EdgeWeight::make(0, 0)),                              // multiply -> effect out
                  Edge::make(const_hnd, mult_hnd, EdgeWeight::make(f32_to_bits(5.0f), 1))};         // const -> multiply (B)
    auto meta = EffectMeta::make("MulBy2", {}, {EffectInput::make("source", 0)}, {EffectOutput::make("result", 0)});
    return EffectDesc::make(meta, list);
}

static void load_multby2() {
    auto [dispatch, rx] = test_setup();
    // let dir = TempDir::new("libfriendship")
    char tmpl[] = "/tmp/libfriendship.XXXXXX";
    const char *dir = mkdtemp(tmpl);
    if (!dir) throw std::runtime_error("mkdtemp failed");
    auto mulby2_desc = create_multby2();

    // Add the temp dir as a search dir
    dispatch.dispatch(OscResMan::AddDir{dir});

    // Write the effect definition to file (serde_json::to_writer)
    std::string mulby2_path = std::string(dir) + "/mulby2.fnd";
    {
        std::ofstream f(mulby2_path, std::ios::binary);
        f << mulby2_desc.to_json_string();
    }
    // Determine the hash of our file
    std::string bytes;
    {
        std::ifstream f(mulby2_path, std::ios::binary);
        std::ostringstream ss;
        ss << f.rdbuf();
        bytes = ss.str();
    }
    auto sha = friendship::sha256(bytes);

    // Create the MulBy2 node (id=1), found on disk by its sha256
    auto mul_hnd = NodeHandle::make(1);
    dispatch.dispatch(OscRouteGraph::AddNode{mul_hnd, EffectId::make("MulBy2", sha, {})});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(mul_hnd, EdgeWeight::make(0, 0))});
    auto const_hnd = NodeHandle::make(2);
    dispatch.dispatch(OscRouteGraph::AddNode{const_hnd, EffectId::make("Constant", std::nullopt, {"primitive:///F32Constant"})});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(const_hnd, mul_hnd, EdgeWeight::make(f32_to_bits(0.5f), 0))});
    dispatch.dispatch(render_range(0, 4, 1));
    ASSERT_EQ_ARR(recv(rx), array({2.5f, 2.5f, 2.5f, 2.5f}));

    // a wrong hash, or a file that is not an effect, finds nothing: dispatch::Error::EffectError(NoMatchingEffect)
    {
        std::ofstream f(std::string(dir) + "/garbage.fnd");
        f << "{not json";
    }
    auto bad = sha;
    bad[0] ^= 1;
    for (auto id : {EffectId::make("MulBy2", bad, {}), EffectId::make("NoSuchEffect", std::nullopt, {})}) {
        bool raised = false;
        try {
            dispatch.dispatch(OscRouteGraph::AddNode{NodeHandle::make(7), id});
        } catch (const friendship::dispatch::Error &e) {
            raised = e.kind == friendship::dispatch::Error::EffectError;
        }
        if (!raised) throw std::runtime_error("expected EffectError(NoMatchingEffect)");
    }
    std::filesystem::remove_all(dir);
}

// ---- the effect files this repository ships (effects/*.fnd, written by effects/make_effects.py) -----------------------
// Every file is found through the host's ResMan by the sha256 of its bytes and instantiated as ONE composite node; the
// same definition is then read back and its nodes placed at top level of a second graph (each of the effect's inputs wired
// to the same source); both graphs render two contiguous ranges from the same external rows and must agree bit for bit.
// (Which bits those are is the parity tests' business: tests/test_hip_parity.py renders composites of the same shapes on
// the HIP engine against the oracle.)
static std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot read " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
static void shipped_effect_files() {
    const std::string dir = (std::filesystem::path(__FILE__).parent_path() / ".." / ".." / "effects").lexically_normal().string();
    struct Src { bool external; uint32_t slot; float value; };   // what feeds input k of the effect
    struct Case { const char *file, *name; std::vector<Src> in; };
    const std::vector<Case> cases = {
        {"partial.fnd", "Partial", {{true, 0, 0}, {false, 0, 0.0123f}, {false, 0, 0.7f}}},
        {"triangle.fnd", "TrianglePartial", {{true, 0, 0}, {false, 0, 0.031f}, {false, 0, 0.4f}}},
        {"envelope.fnd", "Envelope", {{true, 0, 0}, {true, 1, 0}}},
        {"tap.fnd", "Tap", {{true, 1, 0}, {false, 0, 0.5f}, {false, 0, 3.0f}}},
        {"voice4.fnd", "Voice4", {{true, 0, 0}, {false, 0, 0.004f}}},
    };
    for (const Case &cs : cases) {
        const std::string text = slurp(dir + "/" + cs.file);
        const auto sha = friendship::sha256(text);
        const EffectDesc desc = EffectDesc::from_json_string(text);
        if (desc.to_json_string() != text) throw std::runtime_error(std::string(cs.file) + ": not in the canonical wire form");
        if (desc.meta.id.name != cs.name || desc.meta.inputs_.size() != cs.in.size()) throw std::runtime_error(std::string(cs.file) + ": unexpected meta");
        auto [inst, rx_i] = test_setup();
        auto [flat, rx_f] = test_setup();
        const auto const_hnd = NodeHandle::make(5000);
        for (TestDispatch *d : {&inst, &flat}) {
            d->dispatch(OscResMan::AddDir{dir});
            d->dispatch(OscRouteGraph::AddNode{const_hnd, const_id()});
        }
        auto source_edge = [&](const Src &s, NodeHandle to, uint32_t to_slot) {
            return s.external ? Edge::new_from_null(to, EdgeWeight::make(s.slot, to_slot))
                              : Edge::make(const_hnd, to, EdgeWeight::make(f32_to_bits(s.value), to_slot));
        };
        // (a) one composite node, looked up by hash
        const auto hnd = NodeHandle::make(1);
        inst.dispatch(OscRouteGraph::AddNode{hnd, EffectId::make(cs.name, sha, {})});
        for (uint32_t k = 0; k < cs.in.size(); ++k) inst.dispatch(OscRouteGraph::AddEdge{source_edge(cs.in[k], hnd, k)});
        inst.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(hnd, EdgeWeight::make(0, 0))});
        // (b) the definition's own nodes at top level
        const uint32_t off = 100;
        for (auto &hn : desc.adjlist.nodes) flat.dispatch(OscRouteGraph::AddNode{NodeHandle::make(off + hn.first.node_handle), hn.second});
        for (const Edge &e : desc.adjlist.edges) {
            const NodeHandle to = e.to.is_toplevel() ? NodeHandle::toplevel() : NodeHandle::make(off + e.to.node_handle);
            if (e.from.is_toplevel()) {
                if (e.to.is_toplevel()) throw std::runtime_error("input wired straight to output: not in these files");
                flat.dispatch(OscRouteGraph::AddEdge{source_edge(cs.in.at(e.weight.from_slot), to, e.weight.to_slot)});
            } else {
                flat.dispatch(OscRouteGraph::AddEdge{Edge::make(NodeHandle::make(off + e.from.node_handle), to, e.weight)});
            }
        }
        bool any_nonzero = false;
        for (auto range : {std::pair<uint64_t, uint64_t>{0, 64}, {64, 200}}) {
            const size_t n = (size_t)(range.second - range.first);
            std::vector<float> ramp(n), noise(n);
            for (size_t i = 0; i < n; ++i) {
                const uint64_t t = range.first + i;
                ramp[i] = (float)t;
                noise[i] = (float)(int)((t * 2654435761u >> 7) % 2001u) / 1000.0f - 1.0f;
            }
            Array2 got[2];
            int which = 0;
            for (auto pr : {std::pair<TestDispatch *, Channel>{&inst, rx_i}, {&flat, rx_f}}) {
                Jagged2 rows;
                rows.extend(ramp.data(), n);
                rows.extend(noise.data(), n);
                pr.first->dispatch(render_range(range.first, range.second, 1, rows));
                got[which++] = recv(pr.second);
            }
            if (got[0].data.size() != n || got[1].data.size() != n || std::memcmp(got[0].data.data(), got[1].data.data(), n * sizeof(float)) != 0)
                throw std::runtime_error(std::string(cs.file) + ": the instance and its definition placed at top level render different bits");
            for (float v : got[0].data) any_nonzero = any_nonzero || v != 0.0f;
        }
        if (!any_nonzero) throw std::runtime_error(std::string(cs.file) + ": rendered nothing but zeros");
    }
}

// ---- block streaming through the plugin (fr_stream_*; the HIP engine only -- the CPU oracle says "unsupported") --------------
// A voice of 128 file-defined Partial instances under a Sum2 tree, rendered block by block through the resident launch and,
// by a second renderer, through fill_buffer: same bits.
static void block_streaming() {
    const char *lib = std::getenv("FRIENDSHIP_RENDERER_LIB");
    const std::string dir = (std::filesystem::path(__FILE__).parent_path() / ".." / ".." / "effects").lexically_normal().string();
    const auto sha = friendship::sha256(slurp(dir + "/partial.fnd"));
    resman::ResMan res;
    res.add_dir(dir);
    auto partial = routing::Effect::from_id(EffectId::make("Partial", sha, {}), res);
    auto constant = routing::Effect::from_id(const_id(), res);
    auto sum2 = routing::Effect::from_id(sum2_id(), res);
    render::PluginRenderer streamed(lib), plain(lib);
    const uint32_t P = 128;
    for (render::PluginRenderer *r : {&streamed, &plain}) {
        r->on_add_node(NodeHandle::make(1), constant);
        std::vector<uint32_t> level;
        uint32_t next = 2;
        for (uint32_t k = 0; k < P; ++k) {
            const auto h = NodeHandle::make(next++);
            r->on_add_node(h, partial);
            r->on_add_edge(Edge::new_from_null(h, EdgeWeight::make(0, 0)));
            r->on_add_edge(Edge::make(NodeHandle::make(1), h, EdgeWeight::make(f32_to_bits(0.0011f * (float)(k + 1)), 1)));
            r->on_add_edge(Edge::make(NodeHandle::make(1), h, EdgeWeight::make(f32_to_bits(1.0f / (float)(k + 1)), 2)));
            level.push_back(h.node_handle);
        }
        while (level.size() > 1) {
            std::vector<uint32_t> up;
            for (size_t i = 0; i + 1 < level.size(); i += 2) {
                const auto h = NodeHandle::make(next++);
                r->on_add_node(h, sum2);
                r->on_add_edge(Edge::make(NodeHandle::make(level[i]), h, EdgeWeight::make(0, 0)));
                r->on_add_edge(Edge::make(NodeHandle::make(level[i + 1]), h, EdgeWeight::make(0, 1)));
                up.push_back(h.node_handle);
            }
            level.swap(up);
        }
        r->on_add_edge(Edge::new_to_null(NodeHandle::make(level[0]), EdgeWeight::make(0, 0)));
    }
    if (!streamed.stream_begin(1)) {
        if (streamed.backend() == "hip-gfx950") throw std::runtime_error("the HIP engine refused to stream a template voice");
        return;   // the CPU oracle: nothing to stream with
    }
    uint64_t idx = 0;
    for (size_t n : {64u, 17u, 64u, 1u, 40u}) {
        std::vector<float> row(n);
        for (size_t i = 0; i < n; ++i) row[i] = (float)(idx + i);
        Array2 a{1, n, std::vector<float>(n, -1.0f)}, b{1, n, std::vector<float>(n, 0.0f)};
        streamed.stream_block(a, idx, row);
        Jagged2 rows;
        rows.extend(row.data(), n);
        plain.fill_buffer(b, idx, rows);
        if (std::memcmp(a.data.data(), b.data.data(), n * sizeof(float)) != 0) throw std::runtime_error("a streamed block differs from fill_buffer");
        idx += n;
    }
    streamed.stream_end();
}

// The wire shape serde derives for EffectDesc, byte for byte, and its round trip.
static void effect_desc_json() {
    auto desc = create_multby2();
    std::string text = desc.to_json_string();
    const char *want =
        "{\"meta\":{\"id\":{\"name\":\"MulBy2\",\"sha256\":null,\"urls\":[]},\"inputs\":[{\"name\":\"source\",\"channel\":0}],"
        "\"outputs\":[{\"name\":\"result\",\"channel\":0}]},\"adjlist\":{\"nodes\":[[{\"node_handle\":1},{\"name\":\"Multiply\","
        "\"sha256\":null,\"urls\":[\"primitive:///Multiply\"]}],[{\"node_handle\":2},{\"name\":\"Constant\",\"sha256\":null,"
        "\"urls\":[\"primitive:///F32Constant\"]}]],\"edges\":[{\"from\":{\"node_handle\":0},\"to\":{\"node_handle\":1},"
        "\"weight\":{\"from_slot\":0,\"to_slot\":0}},{\"from\":{\"node_handle\":1},\"to\":{\"node_handle\":0},\"weight\":"
        "{\"from_slot\":0,\"to_slot\":0}},{\"from\":{\"node_handle\":2},\"to\":{\"node_handle\":1},\"weight\":"
        "{\"from_slot\":1084227584,\"to_slot\":1}}]}}";
    if (text != want) throw std::runtime_error("unexpected serialisation: " + text);
    auto back = EffectDesc::from_json_string(text);
    if (back.to_json_string() != text) throw std::runtime_error("round trip changed the description");
    // FIPS 180-4 test vector
    auto h = friendship::sha256(std::string("abc"));
    const uint8_t abc[4] = {0xba, 0x78, 0x16, 0xbf};
    if (std::memcmp(h.data(), abc, 4) != 0 || h[31] != 0xad) throw std::runtime_error("sha256 is wrong");
}

// ---- RouteGraph validation (src/routing/routegraph.rs:165-208): host-side, no renderer compute ----
template <class F>
static void expect_rg_error(routegraph::ErrorKind want, F &&f) {
    try {
        f();
    } catch (const dispatch::Error &e) {
        if (e.kind == dispatch::Error::RouteGraphError && e.routegraph_kind && *e.routegraph_kind == want) return;
        throw std::runtime_error(std::string("wrong error: ") + e.what());
    }
    throw std::runtime_error(std::string("expected ") + routegraph::Error::name(want));
}

static void routegraph_validation() {
    auto [dispatch, rx] = test_setup();
    (void)rx;
    auto a = NodeHandle::make(1), b = NodeHandle::make(2), c = NodeHandle::make(3);
    dispatch.dispatch(OscRouteGraph::AddNode{a, sum2_id()});
    dispatch.dispatch(OscRouteGraph::AddNode{b, mult_id()});
    dispatch.dispatch(OscRouteGraph::AddNode{c, const_id()});
    expect_rg_error(routegraph::ErrorKind::NodeExists, [&] { dispatch.dispatch(OscRouteGraph::AddNode{a, sum2_id()}); });
    expect_rg_error(routegraph::ErrorKind::NoSuchNode, [&] {
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(a, NodeHandle::make(9), EdgeWeight::make(0, 0))});
    });
    expect_rg_error(routegraph::ErrorKind::NoSuchSlot, [&] {   // Sum2 has inputs 0 and 1 only
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(c, a, EdgeWeight::make(f32_to_bits(1.f), 2))});
    });
    expect_rg_error(routegraph::ErrorKind::NoSuchSlot, [&] {   // Sum2 has one output
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(a, b, EdgeWeight::make(1, 0))});
    });
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(a, b, EdgeWeight::make(0, 0))});
    expect_rg_error(routegraph::ErrorKind::SlotAlreadyConnected, [&] {
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(c, b, EdgeWeight::make(f32_to_bits(1.f), 0))});
    });
    expect_rg_error(routegraph::ErrorKind::WouldCycle, [&] {
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(b, a, EdgeWeight::make(0, 0))});
    });
    expect_rg_error(routegraph::ErrorKind::WouldCycle, [&] {
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(b, b, EdgeWeight::make(0, 1))});
    });
    expect_rg_error(routegraph::ErrorKind::NodeInUse, [&] { dispatch.dispatch(OscRouteGraph::DelNode{a}); });
    dispatch.dispatch(OscRouteGraph::DelEdge{Edge::make(a, b, EdgeWeight::make(0, 0))});
    dispatch.dispatch(OscRouteGraph::DelNode{a});
    dispatch.dispatch(OscRouteGraph::DelNode{a});   // already deleted: Ok(())
}

// client/chanclient.rs: MpscClient forwards callbacks over a channel that another thread can drain; QueryMeta/QueryId
// reach the client (dispatch.rs:132-145); messages carry the reference's OSC addresses.
// Feedback with the host mirror switched to "the reference as written": AddEdge accepts the edge that closes the loop
// (routegraph.rs:218-237 never refuses it) and the renderer evaluates x = 1 + 0.5 * Delay(x, 2) as RefRenderer's recursion
// would (reference.rs:197-216): 1, 1, 1.5, 1.5, 1.75, 1.75 ...; a loop with no Delay on it is refused at RenderRange.
static void feedback_reference_as_written() {
    auto [dispatch, rx] = test_setup();
    auto c = NodeHandle::make(1), x = NodeHandle::make(2), d = NodeHandle::make(3), m = NodeHandle::make(4);
    dispatch.dispatch(OscRouteGraph::AddNode{c, const_id()});
    dispatch.dispatch(OscRouteGraph::AddNode{x, sum2_id()});
    dispatch.dispatch(OscRouteGraph::AddNode{d, delay_id()});
    dispatch.dispatch(OscRouteGraph::AddNode{m, mult_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(c, x, EdgeWeight::make(f32_to_bits(1.f), 0))});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(m, x, EdgeWeight::make(0, 1))});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(d, m, EdgeWeight::make(0, 0))});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(c, m, EdgeWeight::make(f32_to_bits(0.5f), 1))});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(c, d, EdgeWeight::make(f32_to_bits(2.f), 1))});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(x, EdgeWeight::make(0, 0))});
    expect_rg_error(routegraph::ErrorKind::WouldCycle, [&] {   // the documented behaviour is the default
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(x, d, EdgeWeight::make(0, 0))});
    });
    dispatch.set_reference_as_written(true);
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(x, d, EdgeWeight::make(0, 0))});
    dispatch.dispatch(render_range(0, 6, 1));
    ASSERT_EQ_ARR(recv(rx), array({1.f, 1.f, 1.5f, 1.5f, 1.75f, 1.75f}));
    dispatch.dispatch(render_range(6, 10, 1));
    ASSERT_EQ_ARR(recv(rx), array({1.875f, 1.875f, 1.9375f, 1.9375f}));
    // a second loop with no Delay on it: accepted as an edge, refused when it is rendered (the reference would overflow its
    // stack -- and so would the oracle, which restates it: not run there)
    const char *lib = std::getenv("FRIENDSHIP_RENDERER_LIB");
    if (lib && std::string(lib).find("fr_oracle") != std::string::npos) return;
    auto s1 = NodeHandle::make(5);
    dispatch.dispatch(OscRouteGraph::AddNode{s1, sum2_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(s1, s1, EdgeWeight::make(0, 0))});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(s1, EdgeWeight::make(0, 1))});
    bool refused = false;
    try { dispatch.dispatch(render_range(10, 12, 2)); } catch (const std::exception &) { refused = true; }
    if (!refused) throw std::runtime_error("a delay-free loop rendered");
}

// The dense call (inputs as the reference's Array2) through the C++ plugin wrapper, and the track declaration: out = in0 * in1
// with in1 declared a track -- refused on the device (a track read by something that is not a voice leaf), rendered by the
// oracle (which stores tracks like any input); without the declaration both render it.
static void dense_inputs_and_track_declaration() {
    const char *lib = std::getenv("FRIENDSHIP_RENDERER_LIB");
    const bool oracle = lib && std::string(lib).find("fr_oracle") != std::string::npos;
    render::PluginRenderer r(lib ? lib : "libfriendship_hip.so");
    routing::NodeData mul = routing::Effect::from_id(mult_id(), resman::ResMan());
    r.on_add_node(NodeHandle::make(1), mul);
    r.on_add_edge(Edge::make(NodeHandle::toplevel(), NodeHandle::make(1), EdgeWeight::make(0, 0)));
    r.on_add_edge(Edge::make(NodeHandle::toplevel(), NodeHandle::make(1), EdgeWeight::make(1, 1)));
    r.on_add_edge(Edge::new_to_null(NodeHandle::make(1), EdgeWeight::make(0, 0)));
    Array2 in = Array2::zeros(2, 4), out = Array2::zeros(1, 4);
    for (size_t t = 0; t < 4; ++t) { in.at(0, t) = (float)(t + 1); in.at(1, t) = 0.5f; }
    r.fill_buffer_dense(out, 0, in);
    ASSERT_EQ_ARR(out, array({0.5f, 1.f, 1.5f, 2.f}));
    r.set_track_inputs(1);
    bool refused = false;
    try { r.fill_buffer_dense(out, 4, in); } catch (const std::exception &) { refused = true; }
    if (refused == oracle) throw std::runtime_error(oracle ? "the oracle refused a declared track" : "a track read by a plain node was rendered");
}

static void mpsc_client_and_queries() {
    const char *lib = std::getenv("FRIENDSHIP_RENDERER_LIB");
    auto [client, rx] = friendship::client::MpscClient::make();
    std::unique_ptr<render::Renderer> r = std::make_unique<render::PluginRenderer>(lib);
    Dispatch<std::unique_ptr<render::Renderer>, friendship::client::MpscClient> dispatch(std::move(r), std::move(client));
    auto h = NodeHandle::make(1);
    dispatch.dispatch(OscRouteGraph::AddNode{h, const_id()});
    dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(h, EdgeWeight::make(f32_to_bits(0.25f), 0))});
    dispatch.dispatch(OscRouteGraph::QueryMeta{h});
    dispatch.dispatch(OscRouteGraph::QueryId{h});
    dispatch.dispatch(OscRouteGraph::QueryId{NodeHandle::make(99)});   // unknown handle: only a warning, no message
    dispatch.dispatch(render_range(0, 2, 1));
    auto m1 = rx.recv(), m2 = rx.recv(), m3 = rx.recv();
    if (m1.kind != friendship::client::ClientMessage::NodeMeta || m1.meta.id.name != "F32Constant") throw std::runtime_error("NodeMeta");
    if (m2.kind != friendship::client::ClientMessage::NodeId || !m2.id.is_primitive()) throw std::runtime_error("NodeId");
    if (m3.kind != friendship::client::ClientMessage::AudioRendered || m3.idx != 0) throw std::runtime_error("AudioRendered");
    ASSERT_EQ_ARR(m3.buffer, array({0.25f, 0.25f}));
    if (rx.try_recv()) throw std::runtime_error("unexpected extra message");
    using friendship::dispatch::OscToplevel;
    using friendship::dispatch::osc_address;
    if (osc_address(OscToplevel(OscRouteGraph::Msg(OscRouteGraph::AddNode{h, const_id()}))) != "/routegraph/add_node" ||
        osc_address(OscToplevel(OscRouteGraph::Msg(OscRouteGraph::QueryId{h}))) != "/routegraph/query_id" ||
        osc_address(OscToplevel(OscRenderer::Msg(render_range(0, 1, 1)))) != "/renderer/render" ||
        osc_address(OscToplevel(OscResMan::Msg(OscResMan::AddDir{"x"}))) != "/resman/add_dir")
        throw std::runtime_error("osc_address");
}

// Two GPUs' worth of Dispatch: every rank's Dispatch<HipRenderer, _> is sent the SAME messages (dispatch.rs:111-161) and its
// renderer is told which rank it is -- the reference's message stream needs no change.  Voices mode: rank r's buffer holds
// its block of the rows, the other rows keep the zeros Dispatch allocated (dispatch.rs:149).
static void sharded_dispatch_two_ranks() {
    const char *lib = std::getenv("FRIENDSHIP_RENDERER_LIB");
    std::vector<Array2> got;
    for (uint32_t rank = 0; rank < 2; ++rank) {
        Channel rx = std::make_shared<std::deque<Array2>>();
        auto pr = std::make_unique<render::PluginRenderer>(lib);
        pr->set_shard(rank, 2, FR_SHARD_VOICES);
        auto rows = pr->shard_rows(3);
        if (rows != (rank == 0 ? std::make_pair(0u, 2u) : std::make_pair(2u, 3u))) throw std::runtime_error("shard_rows");
        std::unique_ptr<render::Renderer> r = std::move(pr);
        TestDispatch dispatch(std::move(r), MyClient(rx));
        auto c = NodeHandle::make(1), m = NodeHandle::make(2);
        dispatch.dispatch(OscRouteGraph::AddNode{c, const_id()});
        dispatch.dispatch(OscRouteGraph::AddNode{m, mult_id()});
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(c, m, EdgeWeight::make(f32_to_bits(3.0f), 0))});
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::make(c, m, EdgeWeight::make(f32_to_bits(0.5f), 1))});
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(c, EdgeWeight::make(f32_to_bits(7.0f), 0))});    // row 0 = 7
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(m, EdgeWeight::make(0, 1))});                      // row 1 = 1.5
        dispatch.dispatch(OscRouteGraph::AddEdge{Edge::new_to_null(c, EdgeWeight::make(f32_to_bits(-2.0f), 2))});   // row 2 = -2
        dispatch.dispatch(render_range(0, 2, 3));
        got.push_back(recv(rx));
    }
    ASSERT_EQ_ARR(got[0], (Array2{3, 2, {7.0f, 7.0f, 1.5f, 1.5f, 0.0f, 0.0f}}));
    ASSERT_EQ_ARR(got[1], (Array2{3, 2, {0.0f, 0.0f, 0.0f, 0.0f, -2.0f, -2.0f}}));
}

int main(int argc, char **argv) {
    std::vector<std::pair<const char *, std::function<void()>>> tests = {
        {"sharded_dispatch_two_ranks", sharded_dispatch_two_ranks},
        {"render_zeros", render_zeros}, {"render_const", render_const}, {"render_delay", render_delay},
        {"render_mult", render_mult}, {"render_sum2", render_sum2}, {"render_div", render_div},
        {"render_mod", render_mod}, {"render_min", render_min},
        {"ext_render_passthrough", ext_render_passthrough}, {"ext_render_delay", ext_render_delay},
        {"load_multby2", load_multby2}, {"shipped_effect_files", shipped_effect_files}, {"block_streaming", block_streaming}, {"effect_desc_json", effect_desc_json},
        {"routegraph_validation", routegraph_validation}, {"feedback_reference_as_written", feedback_reference_as_written}, {"dense_inputs_and_track_declaration", dense_inputs_and_track_declaration},
        {"mpsc_client_and_queries", mpsc_client_and_queries}};
    int failed = 0, ran = 0;
    for (auto &t : tests) {
        if (argc > 1 && std::string(argv[1]) != t.first) continue;
        ++ran;
        try {
            t.second();
            std::printf("test %s ... ok\n", t.first);
        } catch (const std::exception &e) {
            std::printf("test %s ... FAILED: %s\n", t.first, e.what());
            ++failed;
        }
    }
    std::printf("test result: %s. %d passed; %d failed\n", failed ? "FAILED" : "ok", ran - failed, failed);
    return failed ? 1 : 0;
}
