// friendship.hpp -- C++ host-side mirror of the reference's routing / render / dispatch surface.
//
// The reference is a Rust crate and no Rust toolchain exists in this environment, so the host side
// above the C ABI is written in C++ with the reference's own names, argument meaning and error
// behaviour, so that host programs and tests read like the reference's (tests/cpp/*.cpp transcribe
// tests/render_prim.rs, tests/ext_input.rs, tests/load_effect.rs).  A Rust host keeps its own
// routing/dispatch code and binds only the C ABI.
//
//   friendship::routing::{NodeHandle, EdgeWeight, Edge, EffectId, EffectMeta, EffectIO, EffectDesc,
//                         PrimitiveEffect, Effect, AdjList, RouteGraph, GraphWatcher}
//        <- src/routing/{routegraph,effect,adjlist,graphwatcher,nullable_int}.rs
//   friendship::render::{Renderer, PluginRenderer, HipRenderer}      <- src/render/renderer.rs
//   friendship::resman::ResMan                                        <- src/resman.rs (in-memory form)
//   friendship::client::Client                                        <- src/client/client.rs
//   friendship::dispatch::{OscRouteGraph, OscRenderer, OscResMan, OscToplevel, Dispatch, Error}
//        <- src/dispatch.rs
#pragma once

#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <variant>
#include <vector>

#include <filesystem>
#include <fstream>
#include <sstream>

#include "../../include/friendship_render.h"
#include "json_sha.hpp"

namespace friendship {

// ndarray::Array2<f32>, row-major [rows, cols] (what Dispatch allocates, dispatch.rs:149)
struct Array2 {
    size_t rows = 0, cols = 0;
    std::vector<float> data;
    static Array2 zeros(size_t r, size_t c) { return Array2{r, c, std::vector<float>(r * c, 0.0f)}; }
    float &at(size_t r, size_t c) { return data[r * cols + c]; }
    float at(size_t r, size_t c) const { return data[r * cols + c]; }
    bool operator==(const Array2 &o) const {   // exact f32 equality, like assert_eq! on arrays
        return rows == o.rows && cols == o.cols &&
               std::memcmp(data.data(), o.data.data(), data.size() * sizeof(float)) == 0;
    }
};

// jagged_array::Jagged2<f32> (+ Jagged2Builder::extend)
struct Jagged2 {
    std::vector<float> data;
    std::vector<uint64_t> offsets{0};
    void extend(std::initializer_list<float> row) { extend(row.begin(), row.size()); }
    void extend(const float *row, size_t n) {
        data.insert(data.end(), row, row + n);
        offsets.push_back(data.size());
    }
    uint32_t len() const { return (uint32_t)(offsets.size() - 1); }
};

namespace routing {

// routegraph.rs:29-36; nullable_int.rs: 0 <=> None
struct NodeHandle {
    uint32_t node_handle = 0;
    static NodeHandle toplevel() { return NodeHandle{0}; }
    static NodeHandle make(uint32_t h) { return NodeHandle{h}; }
    bool is_toplevel() const { return node_handle == 0; }
    bool operator==(const NodeHandle &o) const { return node_handle == o.node_handle; }
    bool operator<(const NodeHandle &o) const { return node_handle < o.node_handle; }
};

// routegraph.rs:20-25
struct EdgeWeight {
    uint32_t from_slot = 0, to_slot = 0;
    static EdgeWeight make(uint32_t from_slot, uint32_t to_slot) { return EdgeWeight{from_slot, to_slot}; }
};

// routegraph.rs:38-44,357-390
struct Edge {
    NodeHandle from, to;
    EdgeWeight weight;
    static Edge new_to_null(NodeHandle from, EdgeWeight w) { return Edge{from, NodeHandle::toplevel(), w}; }
    static Edge new_from_null(NodeHandle to, EdgeWeight w) { return Edge{NodeHandle::toplevel(), to, w}; }
    static Edge make(NodeHandle from, NodeHandle to, EdgeWeight w) { return Edge{from, to, w}; }
    NodeHandle from_full() const { return from; }
    NodeHandle to_full() const { return to; }
    uint32_t to_slot() const { return weight.to_slot; }
    uint32_t from_slot() const { return weight.from_slot; }
    bool operator<(const Edge &o) const {
        return std::tie(from.node_handle, to.node_handle, weight.from_slot, weight.to_slot) <
               std::tie(o.from.node_handle, o.to.node_handle, o.weight.from_slot, o.weight.to_slot);
    }
    bool operator==(const Edge &o) const { return !(*this < o) && !(o < *this); }
    fr_edge c() const { return fr_edge{from.node_handle, to.node_handle, weight.from_slot, weight.to_slot}; }
};

inline uint32_t f32_to_bits(float f) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return b;
}

// effect.rs:86-112, declaration order == FR_PRIM_* of the C ABI
enum class PrimitiveEffect { Delay, F32Constant, Sum2, Multiply, Divide, Modulo, Minimum };

// effect.rs:357-377.  `url` is the full URL text, e.g. "primitive:///Delay".
inline std::optional<PrimitiveEffect> primitive_from_url(const std::string &url) {
    static const std::string scheme = "primitive://";
    if (url.compare(0, scheme.size(), scheme) != 0) return std::nullopt;
    std::string path = url.substr(scheme.size());   // authority is empty in "primitive:///X": path = "/X"
    if (path == "/Delay") return PrimitiveEffect::Delay;
    if (path == "/F32Constant") return PrimitiveEffect::F32Constant;
    if (path == "/Sum2") return PrimitiveEffect::Sum2;
    if (path == "/Multiply") return PrimitiveEffect::Multiply;
    if (path == "/Divide") return PrimitiveEffect::Divide;
    if (path == "/Modulo") return PrimitiveEffect::Modulo;
    if (path == "/Minimum") return PrimitiveEffect::Minimum;
    return std::nullopt;   // "Unrecognized primitive effect"
}

// effect.rs:27-39,226-256
struct EffectId {
    std::string name;
    std::optional<std::array<uint8_t, 32>> sha256;
    std::set<std::string> urls;
    static EffectId make(std::string name, std::optional<std::array<uint8_t, 32>> sha256, std::vector<std::string> urls) {
        EffectId id;
        id.name = std::move(name);
        id.sha256 = sha256;
        id.urls.insert(urls.begin(), urls.end());
        return id;
    }
    bool is_primitive() const {
        return urls.size() == 1 && urls.begin()->compare(0, 10, "primitive:") == 0;
    }
    std::optional<std::string> get_primitive_url() const {
        return is_primitive() ? std::optional<std::string>(*urls.begin()) : std::nullopt;
    }
};

// effect.rs:67-74,339-355
struct EffectIO {
    std::string name;
    uint8_t channel = 0;
    static EffectIO make(std::string name, uint8_t channel) { return EffectIO{std::move(name), channel}; }
};
using EffectInput = EffectIO;
using EffectOutput = EffectIO;

// effect.rs:59-65,284-337
struct EffectMeta {
    EffectId id;
    std::vector<EffectInput> inputs_;
    std::vector<EffectOutput> outputs_;
    static EffectMeta make(std::string name, std::vector<std::string> urls, std::vector<EffectInput> in, std::vector<EffectOutput> out) {
        return EffectMeta{EffectId::make(std::move(name), std::nullopt, std::move(urls)), std::move(in), std::move(out)};
    }
    std::optional<PrimitiveEffect> prim_effect() const {
        auto u = id.get_primitive_url();
        return u ? primitive_from_url(*u) : std::nullopt;
    }
    // number of valid input slots; primitives have hard-wired arity (effect.rs:297-314)
    uint64_t n_inputs() const {
        auto p = prim_effect();
        if (!p) return inputs_.size();
        return *p == PrimitiveEffect::F32Constant ? 0 : 2;
    }
    // F32Constant exposes one output per f32 bit pattern except 0xFFFFFFFF (effect.rs:390-417)
    uint64_t n_outputs() const {
        auto p = prim_effect();
        if (!p) return outputs_.size();
        return *p == PrimitiveEffect::F32Constant ? 0xFFFFFFFFull : 1;
    }
    bool is_valid_input(uint32_t slot) const { return slot < n_inputs(); }
    bool is_valid_output(uint32_t slot) const { return slot < n_outputs(); }
};

struct AdjList;   // adjlist.rs:11-15
class RouteGraph;
class Effect;
using NodeData = std::shared_ptr<const Effect>;   // routegraph.rs:27 (Rc<Effect>)

namespace effect {
struct NoMatchingEffect : std::runtime_error {   // effect.rs:18-22
    EffectId id;
    explicit NoMatchingEffect(EffectId i) : std::runtime_error("NoMatchingEffect(" + i.name + ")"), id(std::move(i)) {}
};
}  // namespace effect

namespace routegraph {
enum class ErrorKind { WouldCycle, NodeInUse, NodeExists, SlotAlreadyConnected, NoSuchNode, NoSuchSlot, EffectError };
struct Error : std::runtime_error {   // routegraph.rs:46-62
    ErrorKind kind;
    explicit Error(ErrorKind k) : std::runtime_error(name(k)), kind(k) {}
    static const char *name(ErrorKind k) {
        switch (k) {
        case ErrorKind::WouldCycle: return "WouldCycle";
        case ErrorKind::NodeInUse: return "NodeInUse";
        case ErrorKind::NodeExists: return "NodeExists";
        case ErrorKind::SlotAlreadyConnected: return "SlotAlreadyConnected";
        case ErrorKind::NoSuchNode: return "NoSuchNode";
        case ErrorKind::NoSuchSlot: return "NoSuchSlot";
        default: return "EffectError";
        }
    }
};
}  // namespace routegraph

// routegraph.rs:69-327
class RouteGraph {
    struct Node {
        std::set<Edge> outbound, inbound;
        NodeData node_data;   // null for the toplevel I/O node
    };
    std::map<NodeHandle, Node> nodes_;

public:
    RouteGraph() { nodes_[NodeHandle::toplevel()] = Node{}; }

    // What add_edge does about an edge that closes a dependency cycle.
    //   Documented (default): WouldCycle, the behaviour routegraph.rs documents (its Error::WouldCycle, its tests' intent).
    //   AsWritten: the edge is accepted, as the reference's code does -- `is_edge_reachable` (routegraph.rs:218-237) has no
    //     base case that returns true, so its WouldCycle never fires.  RefRenderer then evaluates the loop by recursion
    //     (reference.rs:197-216), which ends iff every trip round passes a Delay of >= 1 frames; this engine renders the
    //     same samples for loops closed through a constant Delay and answers FR_ERR_CYCLE at fill_buffer where the
    //     reference would recurse forever (feedback).
    enum class CyclePolicy { Documented, AsWritten };
    void set_cycle_policy(CyclePolicy p) { cycle_policy_ = p; }
    CyclePolicy cycle_policy() const { return cycle_policy_; }

    std::vector<std::pair<NodeHandle, NodeData>> iter_nodes() const {
        std::vector<std::pair<NodeHandle, NodeData>> v;
        for (auto &kv : nodes_) if (kv.second.node_data) v.emplace_back(kv.first, kv.second.node_data);
        return v;
    }
    std::vector<Edge> iter_edges() const {
        std::vector<Edge> v;
        for (auto &kv : nodes_) v.insert(v.end(), kv.second.outbound.begin(), kv.second.outbound.end());
        return v;
    }
    // edges that point into outputs / come from inputs (routegraph.rs:131-139)
    const std::set<Edge> &iter_outbound_edges() const { return nodes_.at(NodeHandle::toplevel()).inbound; }
    const std::set<Edge> &iter_inbound_edges() const { return nodes_.at(NodeHandle::toplevel()).outbound; }
    NodeData get_data(NodeHandle h) const {
        auto it = nodes_.find(h);
        return it == nodes_.end() ? nullptr : it->second.node_data;
    }
    std::vector<Edge> iter_edges_to(NodeHandle h) const {
        auto it = nodes_.find(h);
        return it == nodes_.end() ? std::vector<Edge>{} : std::vector<Edge>(it->second.inbound.begin(), it->second.inbound.end());
    }

    void add_node(NodeHandle handle, NodeData data) {   // routegraph.rs:153-162
        if (nodes_.count(handle)) throw routegraph::Error(routegraph::ErrorKind::NodeExists);
        Node n;
        n.node_data = std::move(data);
        nodes_[handle] = std::move(n);
    }

    void add_edge(const Edge &edge);   // routegraph.rs:165-208 (below, needs Effect)

    void del_node(NodeHandle h) {   // routegraph.rs:263-277
        auto it = nodes_.find(h);
        if (it == nodes_.end()) return;
        if (!it->second.outbound.empty() || !it->second.inbound.empty())
            throw routegraph::Error(routegraph::ErrorKind::NodeInUse);
        nodes_.erase(it);
    }
    void del_edge(const Edge &e) {   // routegraph.rs:278-285
        auto f = nodes_.find(e.from_full());
        if (f != nodes_.end()) f->second.outbound.erase(e);
        auto t = nodes_.find(e.to_full());
        if (t != nodes_.end()) t->second.inbound.erase(e);
    }

    // True if a signal entering the toplevel at `in_slot` can reach toplevel output `out_slot`
    // (routegraph.rs:245-262).
    bool are_slots_connected(uint32_t in_slot, uint32_t out_slot) const;

    AdjList to_adjlist() const;

private:
    CyclePolicy cycle_policy_ = CyclePolicy::Documented;
    // Is there a directed path from node `at`, entered through input slot `at_slot`, to node `target`
    // arriving so that it drives target's output slot `target_out`?  This is what
    // `is_edge_reachable(&edge, &edge)` (routegraph.rs:218-237) is documented to decide.  NOTE: the
    // reference's implementation has no base case and can never return true, so the reference as
    // written never raises WouldCycle; this mirror implements the documented intent.
    bool reaches(NodeHandle at, uint32_t at_slot, NodeHandle target, uint32_t target_out,
                 std::set<std::pair<uint32_t, uint32_t>> &seen) const;
};

// adjlist.rs:11-15
struct AdjList {
    std::vector<std::pair<NodeHandle, EffectId>> nodes;
    std::vector<Edge> edges;
};

// effect.rs:44-48,258-282.  to_json()/from_json() produce and accept the wire shape serde derives for the
// reference's structs: {"meta":{"id":{"name","sha256","urls"},"inputs":[{"name","channel"}],
// "outputs":[...]},"adjlist":{"nodes":[[{"node_handle":n},{id}]...],"edges":[{"from":{"node_handle":n},
// "to":{...},"weight":{"from_slot","to_slot"}}]}}; NodeHandle's NullableInt is a plain integer, 0 = null
// (nullable_int.rs:88-102); sha256 is null or an array of 32 integers.
struct EffectDesc {
    EffectMeta meta;
    AdjList adjlist;
    static EffectDesc make(EffectMeta m, AdjList a) { return EffectDesc{std::move(m), std::move(a)}; }

    static json::Value id_to_json(const EffectId &id) {
        json::Value sha = json::Value::null();
        if (id.sha256) {
            sha = json::Value::array();
            for (uint8_t b : *id.sha256) sha.a->push_back(json::Value::integer(b));
        }
        json::Value urls = json::Value::array();
        for (auto &u : id.urls) urls.a->push_back(json::Value::string(u));
        return json::Value::object({{"name", json::Value::string(id.name)}, {"sha256", sha}, {"urls", urls}});
    }
    static EffectId id_from_json(const json::Value &v) {
        EffectId id;
        id.name = v.at("name").str();
        const json::Value &sha = v.at("sha256");
        if (!sha.is_null()) {
            if (sha.arr().size() != 32) throw std::runtime_error("sha256 must have 32 entries");
            std::array<uint8_t, 32> a{};
            for (size_t i = 0; i < 32; ++i) a[i] = (uint8_t)sha.arr()[i].u64();
            id.sha256 = a;
        }
        for (auto &u : v.at("urls").arr()) id.urls.insert(u.str());
        return id;
    }
    static json::Value handle_to_json(NodeHandle h) {
        return json::Value::object({{"node_handle", json::Value::integer(h.node_handle)}});
    }
    static NodeHandle handle_from_json(const json::Value &v) { return NodeHandle::make((uint32_t)v.at("node_handle").u64()); }

    json::Value to_json() const {
        auto ios = [](const std::vector<EffectIO> &v) {
            json::Value a = json::Value::array();
            for (auto &io : v)
                a.a->push_back(json::Value::object({{"name", json::Value::string(io.name)}, {"channel", json::Value::integer(io.channel)}}));
            return a;
        };
        json::Value nodes = json::Value::array(), edges = json::Value::array();
        for (auto &hn : adjlist.nodes) nodes.a->push_back(json::Value::array({handle_to_json(hn.first), id_to_json(hn.second)}));
        for (auto &e : adjlist.edges)
            edges.a->push_back(json::Value::object(
                {{"from", handle_to_json(e.from)}, {"to", handle_to_json(e.to)},
                 {"weight", json::Value::object({{"from_slot", json::Value::integer(e.weight.from_slot)},
                                                 {"to_slot", json::Value::integer(e.weight.to_slot)}})}}));
        return json::Value::object(
            {{"meta", json::Value::object({{"id", id_to_json(meta.id)}, {"inputs", ios(meta.inputs_)}, {"outputs", ios(meta.outputs_)}})},
             {"adjlist", json::Value::object({{"nodes", nodes}, {"edges", edges}})}});
    }
    std::string to_json_string() const { return json::to_string(to_json()); }   // serde_json::to_writer

    static EffectDesc from_json(const json::Value &v) {
        EffectDesc d;
        const json::Value &m = v.at("meta");
        d.meta.id = id_from_json(m.at("id"));
        auto ios = [](const json::Value &a) {
            std::vector<EffectIO> out;
            for (auto &io : a.arr()) out.push_back(EffectIO::make(io.at("name").str(), (uint8_t)io.at("channel").u64()));
            return out;
        };
        d.meta.inputs_ = ios(m.at("inputs"));
        d.meta.outputs_ = ios(m.at("outputs"));
        const json::Value &adj = v.at("adjlist");
        for (auto &n : adj.at("nodes").arr()) {
            if (n.arr().size() != 2) throw std::runtime_error("adjlist node must be a [handle, id] pair");
            d.adjlist.nodes.emplace_back(handle_from_json(n.arr()[0]), id_from_json(n.arr()[1]));
        }
        for (auto &e : adj.at("edges").arr()) {
            const json::Value &w = e.at("weight");
            d.adjlist.edges.push_back(Edge::make(handle_from_json(e.at("from")), handle_from_json(e.at("to")),
                                                 EdgeWeight::make((uint32_t)w.at("from_slot").u64(), (uint32_t)w.at("to_slot").u64())));
        }
        return d;
    }
    static EffectDesc from_json_string(const std::string &text) { return from_json(json::parse(text)); }

    // effect.rs:272-281: an id without a hash gets the sha256 of the description's own serialisation
    void update_id() {
        if (!meta.id.sha256) meta.id.sha256 = sha256(to_json_string());
    }
};

}  // namespace routing

namespace resman {
// src/resman.rs: a list of search directories plus a sha256 -> path cache.  find_effect() yields every
// candidate file for an id -- the cached path for its sha256 first, then every regular file of every
// directory -- filtered by the sha256 of the file's bytes when the id carries one (resman.rs:39-60).
// add_desc() additionally registers in-memory descriptions (not in the reference; handy for hosts that
// build effects programmatically).
class ResMan {
    std::vector<std::string> dirs_;
    std::vector<routing::EffectDesc> descs_;
    mutable std::map<std::array<uint8_t, 32>, std::string> sha256_to_path_;   // ResCache (resman.rs:25-28,99-108)

    static bool read_file(const std::string &path, std::string &out) {
        std::ifstream f(path, std::ios::binary);
        if (!f) return false;
        std::ostringstream ss;
        ss << f.rdbuf();
        out = ss.str();
        return true;
    }

public:
    struct Candidate { std::string path; std::string text; };

    void add_dir(std::string dir) { dirs_.push_back(std::move(dir)); }
    void add_desc(routing::EffectDesc d) { descs_.push_back(std::move(d)); }
    const std::vector<std::string> &dirs() const { return dirs_; }
    const std::vector<routing::EffectDesc> &descs() const { return descs_; }

    std::vector<Candidate> find_effect(const routing::EffectId &id) const {
        std::vector<std::string> paths;
        if (id.sha256) {   // iter_all_files: the cached path is visited first (and possibly again below)
            auto it = sha256_to_path_.find(*id.sha256);
            if (it != sha256_to_path_.end()) paths.push_back(it->second);
        }
        for (auto &d : dirs_) {
            std::error_code ec;
            std::filesystem::directory_iterator di(d, ec), end;
            if (ec) continue;   // "ResMan: Failed to read directory"
            for (; di != end; di.increment(ec)) {
                if (ec) break;
                if (di->is_regular_file(ec)) paths.push_back(di->path().string());
            }
        }
        std::vector<Candidate> out;
        for (auto &p : paths) {
            Candidate c{p, {}};
            if (!read_file(p, c.text)) continue;
            if (id.sha256) {
                auto h = sha256(c.text);
                sha256_to_path_[h] = p;   // notify_sha256
                if (h != *id.sha256) continue;
            }
            out.push_back(std::move(c));
        }
        return out;
    }
};
}  // namespace resman

namespace routing {

// effect.rs:50-57,76-82,119-224
class Effect {
    EffectMeta meta_;
    std::variant<RouteGraph, PrimitiveEffect> data_;

public:
    Effect(EffectMeta m, std::variant<RouteGraph, PrimitiveEffect> d) : meta_(std::move(m)), data_(std::move(d)) {}
    const EffectId &id() const { return meta_.id; }
    const EffectMeta &meta() const { return meta_; }
    bool is_primitive() const { return data_.index() == 1; }
    PrimitiveEffect primitive() const { return std::get<1>(data_); }
    const RouteGraph &graph() const { return std::get<0>(data_); }
    bool are_slots_connected(uint32_t from_slot, uint32_t to_slot) const {   // effect.rs:120-126
        return is_primitive() ? true : graph().are_slots_connected(from_slot, to_slot);
    }

    static RouteGraph graph_from_adjlist(const AdjList &adj, const resman::ResMan &res);   // routegraph.rs:305-326

    // effect.rs:135-220
    static NodeData from_id(const EffectId &id, const resman::ResMan &res) {
        auto url = id.get_primitive_url();
        auto prim = url ? primitive_from_url(*url) : std::nullopt;
        if (prim && !id.sha256) {
            EffectMeta m;
            m.id = id;   // primitive effects have undocumented I/O: empty lists (effect.rs:145-147)
            return std::make_shared<const Effect>(std::move(m), *prim);
        }
        // candidates: files found by the ResMan (parsed here, like serde_json::from_reader at effect.rs:160),
        // then descriptions registered in memory
        std::vector<EffectDesc> cands;
        for (auto &c : res.find_effect(id)) {
            try {
                cands.push_back(EffectDesc::from_json_string(c.text));
            } catch (const std::exception &) {
                // "Unable to deserialize EffectDesc": skip the file (effect.rs:213-215)
            }
        }
        for (auto &d : res.descs())
            if (!id.sha256 || !d.meta.id.sha256 || *id.sha256 == *d.meta.id.sha256) cands.push_back(d);
        for (EffectDesc &cand : cands) {
            if (cand.meta.id.name != id.name) continue;
            cand.update_id();
            const EffectDesc *desc = &cand;
            try {
                RouteGraph graph = graph_from_adjlist(desc->adjlist, res);
                // all outputs driven, exactly once each, 0..n (effect.rs:168-175)
                std::vector<uint32_t> real_outputs;
                for (auto &e : graph.iter_outbound_edges()) real_outputs.push_back(e.to_slot());
                std::sort(real_outputs.begin(), real_outputs.end());
                bool outputs_driven = real_outputs.size() == desc->meta.outputs_.size();
                for (size_t i = 0; outputs_driven && i < real_outputs.size(); ++i) outputs_driven = real_outputs[i] == i;
                // every input edge is declared (effect.rs:179-188)
                bool inputs_valid = true;
                for (auto &e : graph.iter_inbound_edges()) inputs_valid = inputs_valid && e.from_slot() < desc->meta.inputs_.size();
                // every sub-node has all its inputs driven (effect.rs:189-195)
                bool subnodes_driven = true;
                for (auto &hn : graph.iter_nodes()) {
                    std::vector<uint32_t> driven;
                    for (auto &e : graph.iter_edges_to(hn.first)) driven.push_back(e.to_slot());
                    std::sort(driven.begin(), driven.end());
                    uint64_t want = hn.second->meta().n_inputs();
                    bool ok = driven.size() == want;
                    for (size_t i = 0; ok && i < driven.size(); ++i) ok = driven[i] == i;
                    subnodes_driven = subnodes_driven && ok;
                }
                if (inputs_valid && outputs_driven && subnodes_driven)
                    return std::make_shared<const Effect>(desc->meta, std::move(graph));
            } catch (const routegraph::Error &) {
                // "RouteGraph::from_adjlist failed": try the next candidate (effect.rs:207)
            } catch (const effect::NoMatchingEffect &) {
            }
        }
        throw effect::NoMatchingEffect(id);
    }
};

inline RouteGraph Effect::graph_from_adjlist(const AdjList &adj, const resman::ResMan &res) {
    RouteGraph g;
    for (auto &hn : adj.nodes) g.add_node(hn.first, Effect::from_id(hn.second, res));
    for (auto &e : adj.edges) g.add_edge(e);
    return g;
}

inline void RouteGraph::add_edge(const Edge &edge) {
    using routegraph::Error;
    using routegraph::ErrorKind;
    auto to = nodes_.find(edge.to_full());
    if (to == nodes_.end()) throw Error(ErrorKind::NoSuchNode);
    for (auto &in_edge : to->second.inbound)
        if (in_edge.to_slot() == edge.to_slot()) throw Error(ErrorKind::SlotAlreadyConnected);
    if (to->second.node_data && !to->second.node_data->meta().is_valid_input(edge.to_slot()))
        throw Error(ErrorKind::NoSuchSlot);
    auto from = nodes_.find(edge.from_full());
    if (from == nodes_.end()) throw Error(ErrorKind::NoSuchNode);
    if (from->second.node_data && !from->second.node_data->meta().is_valid_output(edge.from_slot()))
        throw Error(ErrorKind::NoSuchSlot);
    if (cycle_policy_ == CyclePolicy::Documented && !edge.to_full().is_toplevel() && !edge.from_full().is_toplevel()) {
        std::set<std::pair<uint32_t, uint32_t>> seen;
        if (reaches(edge.to_full(), edge.to_slot(), edge.from_full(), edge.from_slot(), seen))
            throw Error(ErrorKind::WouldCycle);
    }
    nodes_[edge.from_full()].outbound.insert(edge);
    nodes_[edge.to_full()].inbound.insert(edge);
}

inline bool RouteGraph::reaches(NodeHandle at, uint32_t at_slot, NodeHandle target, uint32_t target_out,
                                std::set<std::pair<uint32_t, uint32_t>> &seen) const {
    if (at.is_toplevel()) return false;
    if (!seen.insert({at.node_handle, at_slot}).second) return false;
    auto it = nodes_.find(at);
    if (it == nodes_.end() || !it->second.node_data) return false;
    const Effect &eff = *it->second.node_data;
    if (at == target && eff.are_slots_connected(at_slot, target_out)) return true;
    for (auto &out : it->second.outbound)
        if (eff.are_slots_connected(at_slot, out.from_slot()) && reaches(out.to_full(), out.to_slot(), target, target_out, seen))
            return true;
    return false;
}

inline bool RouteGraph::are_slots_connected(uint32_t in_slot, uint32_t out_slot) const {
    // depth-first from every edge leaving toplevel input `in_slot`; arrive at toplevel output `out_slot`
    std::set<std::pair<uint32_t, uint32_t>> seen;
    std::vector<Edge> stack;
    for (auto &e : iter_inbound_edges()) if (e.from_slot() == in_slot) stack.push_back(e);
    while (!stack.empty()) {
        Edge e = stack.back();
        stack.pop_back();
        if (e.to_full().is_toplevel()) {
            if (e.to_slot() == out_slot) return true;
            continue;
        }
        if (!seen.insert({e.to_full().node_handle, e.to_slot()}).second) continue;
        auto it = nodes_.find(e.to_full());
        if (it == nodes_.end() || !it->second.node_data) continue;
        for (auto &out : it->second.outbound)
            if (it->second.node_data->are_slots_connected(e.to_slot(), out.from_slot())) stack.push_back(out);
    }
    return false;
}

inline AdjList RouteGraph::to_adjlist() const {   // routegraph.rs:287-304
    AdjList a;
    for (auto &kv : nodes_) {
        if (kv.second.node_data) a.nodes.emplace_back(kv.first, kv.second.node_data->id());
        a.edges.insert(a.edges.end(), kv.second.outbound.begin(), kv.second.outbound.end());
    }
    return a;
}

// graphwatcher.rs:4-9
struct GraphWatcher {
    virtual void on_add_node(const NodeHandle &node, const NodeData &data) = 0;
    virtual void on_del_node(const NodeHandle &node) = 0;
    virtual void on_add_edge(const Edge &edge) = 0;
    virtual void on_del_edge(const Edge &edge) = 0;
    virtual ~GraphWatcher() = default;
};

}  // namespace routing

namespace render {

// renderer.rs:6-17
struct Renderer : routing::GraphWatcher {
    virtual void fill_buffer(Array2 &buff, uint64_t idx, const Jagged2 &inputs) = 0;
};

// What the reference's trait methods do on a contract violation is panic!; here that is this exception.
struct Panic : std::runtime_error {
    fr_status status;
    Panic(fr_status s, const std::string &m) : std::runtime_error(m), status(s) {}
};

// A renderer living in a shared library that exports include/friendship_render.h.
class PluginRenderer : public Renderer {
protected:
    void *dl_ = nullptr;
    fr_renderer *h_ = nullptr;
    size_t stream_slots_ = 0;            // rows of the stream that is open (fr_stream_block takes no slot count)
    struct Api {
        decltype(&fr_renderer_create) create;
        decltype(&fr_renderer_destroy) destroy;
        decltype(&fr_on_add_node) add_node;
        decltype(&fr_on_del_node) del_node;
        decltype(&fr_on_add_edge) add_edge;
        decltype(&fr_on_del_edge) del_edge;
        decltype(&fr_fill_buffer) fill;
        decltype(&fr_last_error) last_error;
        decltype(&fr_status_string) status_string;
        decltype(&fr_backend_name) backend_name;
        decltype(&fr_set_shard) set_shard;
        decltype(&fr_shard_rows) shard_rows;
        decltype(&fr_stream_begin) stream_begin;
        decltype(&fr_stream_block) stream_block;
        decltype(&fr_stream_end) stream_end;
        decltype(&fr_set_track_inputs) set_track_inputs;
        decltype(&fr_fill_buffer_dense) fill_dense;
    } api_{};

    template <class T>
    void sym(T &fn, const char *name) {
        fn = (T)dlsym(dl_, name);
        if (!fn) throw std::runtime_error(std::string("renderer plugin lacks symbol ") + name);
    }
    void check(fr_status s) const {
        if (s != FR_OK) throw Panic(s, std::string(api_.status_string(s)) + ": " + api_.last_error(h_));
    }

    // Effect -> fr_effect tree (kept alive for the duration of one on_add_node call)
    struct CEffect {
        fr_effect c{};
        std::vector<uint32_t> handles;
        std::vector<std::unique_ptr<CEffect>> children;
        std::vector<const fr_effect *> child_ptrs;
        std::vector<fr_edge> edges;
    };
    static std::unique_ptr<CEffect> lower(const routing::Effect &e) {
        auto ce = std::make_unique<CEffect>();
        if (e.is_primitive()) {
            ce->c.kind = (int32_t)e.primitive();
            return ce;
        }
        ce->c.kind = FR_EFFECT_GRAPH;
        for (auto &hn : e.graph().iter_nodes()) {
            ce->handles.push_back(hn.first.node_handle);
            ce->children.push_back(lower(*hn.second));
            ce->child_ptrs.push_back(&ce->children.back()->c);
        }
        for (auto &ed : e.graph().iter_edges()) ce->edges.push_back(ed.c());
        ce->c.n_nodes = (uint32_t)ce->handles.size();
        ce->c.node_handles = ce->handles.data();
        ce->c.node_effects = ce->child_ptrs.data();
        ce->c.n_edges = (uint32_t)ce->edges.size();
        ce->c.edges = ce->edges.data();
        return ce;
    }

public:
    // `semantics`: FR_SEMANTICS_REFERENCE (RefRenderer, the default) or FR_SEMANTICS_SPARKLE; `history_frames`: 0 = keep
    // every input sample since the last seek (the reference); `flags`: FR_CONFIG_SYNC_COMPILE for deterministic plans.
    explicit PluginRenderer(const std::string &library_path, int mode = FR_MODE_AUTO, int device = -1,
                            int semantics = FR_SEMANTICS_REFERENCE, uint64_t history_frames = 0, uint32_t flags = 0) {
        dl_ = dlopen(library_path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!dl_) throw std::runtime_error(std::string("cannot load renderer library: ") + dlerror());
        sym(api_.create, "fr_renderer_create");
        sym(api_.destroy, "fr_renderer_destroy");
        sym(api_.add_node, "fr_on_add_node");
        sym(api_.del_node, "fr_on_del_node");
        sym(api_.add_edge, "fr_on_add_edge");
        sym(api_.del_edge, "fr_on_del_edge");
        sym(api_.fill, "fr_fill_buffer");
        sym(api_.last_error, "fr_last_error");
        sym(api_.status_string, "fr_status_string");
        sym(api_.backend_name, "fr_backend_name");
        sym(api_.set_shard, "fr_set_shard");
        sym(api_.shard_rows, "fr_shard_rows");
        sym(api_.stream_begin, "fr_stream_begin");
        sym(api_.stream_block, "fr_stream_block");
        sym(api_.stream_end, "fr_stream_end");
        sym(api_.set_track_inputs, "fr_set_track_inputs");
        sym(api_.fill_dense, "fr_fill_buffer_dense");
        fr_config cfg{FR_ABI_VERSION, device, mode, flags, semantics, 0, history_frames};
        fr_status s = api_.create(&cfg, &h_);
        if (s != FR_OK) {
            std::string what = api_.status_string(s);
            dlclose(dl_);
            throw Panic(s, "fr_renderer_create: " + what);
        }
    }
    PluginRenderer(const PluginRenderer &) = delete;
    PluginRenderer &operator=(const PluginRenderer &) = delete;
    ~PluginRenderer() override {
        if (h_) api_.destroy(h_);
        if (dl_) dlclose(dl_);
    }
    std::string backend() const { return api_.backend_name(); }

    // One process (and one renderer) per GPU: this renderer becomes rank `rank` of `world` (friendship_render.h
    // fr_set_shard).  Every rank is sent the same RouteGraph messages and the same RenderRange; it fills the rows
    // shard_rows() names (rank 0 all of them with FR_SHARD_GATHER).  `rccl_id`: the bytes of fr_comm_unique_id from
    // rank 0; `comm`: a host transport instead.  Neither is needed for FR_SHARD_VOICES without gathering.
    void set_shard(uint32_t rank, uint32_t world, int mode = FR_SHARD_VOICES, uint32_t flags = 0, const uint8_t *rccl_id = nullptr,
                   const fr_comm *comm = nullptr) {
        fr_shard sh{rank, world, mode, flags, rccl_id, comm};
        check(api_.set_shard(h_, &sh));
    }
    std::pair<uint32_t, uint32_t> shard_rows(uint32_t n_slots) const {
        uint32_t lo = 0, hi = 0;
        api_.shard_rows(h_, n_slots, &lo, &hi);
        return {lo, hi};
    }

    // Block streaming (friendship_render.h fr_stream_*): a real-time host renders 1..64-frame blocks through one resident launch.
    // stream_begin() returns false where the plugin or the graph cannot be served that way (render with fill_buffer then);
    // any other call on the renderer closes the stream.
    bool stream_begin(uint32_t n_slots) {
        fr_status s = api_.stream_begin(h_, n_slots);
        if (s == FR_ERR_UNSUPPORTED) return false;
        check(s);
        stream_slots_ = n_slots;
        return true;
    }
    void stream_block(Array2 &buff, uint64_t idx, const std::vector<float> &row) {
        // (the C call takes no slot count: it writes the stream's rows, so the buffer is checked here)
        if (buff.rows != stream_slots_ || buff.data.size() != (size_t)buff.rows * buff.cols)
            throw std::invalid_argument("stream_block: the buffer must have the " + std::to_string(stream_slots_) + " rows the stream was begun with");
        check(api_.stream_block(h_, buff.data.data(), buff.cols, idx, row.data(), row.size()));
    }
    void stream_end() { check(api_.stream_end(h_)); }

    // Control-rate tracks (friendship_render.h): input slots >= first_slot are read in place by the voices of the call that supplies
    // them and never stored; fill_buffer_dense takes the inputs in the reference's own shape, an Array2 with one row per slot
    // (renderer.rs:16, reference.rs:66-74).
    void set_track_inputs(uint32_t first_slot) { check(api_.set_track_inputs(h_, first_slot)); }
    void fill_buffer_dense(Array2 &buff, uint64_t idx, const Array2 &inputs) {
        if (inputs.rows != 0 && inputs.cols != buff.cols) throw Panic(FR_ERR_INVALID_ARG, "fill_buffer_dense: inputs must have the buffer's frame count");
        check(api_.fill_dense(h_, buff.data.data(), (uint32_t)buff.rows, buff.cols, idx, inputs.data.data(), (uint32_t)inputs.rows));
    }

    void on_add_node(const routing::NodeHandle &node, const routing::NodeData &data) override {
        auto ce = lower(*data);
        check(api_.add_node(h_, node.node_handle, &ce->c));
    }
    void on_del_node(const routing::NodeHandle &node) override { check(api_.del_node(h_, node.node_handle)); }
    void on_add_edge(const routing::Edge &edge) override {
        fr_edge e = edge.c();
        check(api_.add_edge(h_, &e));
    }
    void on_del_edge(const routing::Edge &edge) override {
        fr_edge e = edge.c();
        check(api_.del_edge(h_, &e));
    }
    void fill_buffer(Array2 &buff, uint64_t idx, const Jagged2 &inputs) override {
        check(api_.fill(h_, buff.data.data(), (uint32_t)buff.rows, buff.cols, idx, inputs.data.data(),
                        inputs.offsets.data(), inputs.len()));
    }
};

// The MI355X renderer: `HipRenderer::default()` plays the role `SparkleRenderer::default()` plays in the
// reference's tests.  Library path: $FRIENDSHIP_HIP_LIB, else libfriendship_hip.so beside this package.
class HipRenderer : public PluginRenderer {
public:
    static std::string default_path() {
        if (const char *p = std::getenv("FRIENDSHIP_HIP_LIB")) return p;
        return "libfriendship_hip.so";
    }
    explicit HipRenderer(int mode = FR_MODE_AUTO, int device = -1) : PluginRenderer(default_path(), mode, device) {}
};

}  // namespace render

namespace client {
// client.rs:8-15
struct Client {
    virtual void audio_rendered(Array2 buffer, uint64_t idx) { (void)buffer; (void)idx; }
    virtual void node_meta(const routing::NodeHandle &, const routing::EffectMeta &) {}
    virtual void node_id(const routing::NodeHandle &, const routing::EffectId &) {}
    virtual ~Client() = default;
};
// chanclient.rs:11-50: every callback becomes a message on a thread-safe channel (std::sync::mpsc there; a
// mutex-guarded queue with a condition variable here).  MpscClient::make() returns the client and its receiver.
struct ClientMessage {
    enum Kind { AudioRendered, NodeMeta, NodeId } kind;
    Array2 buffer;                 // AudioRendered
    uint64_t idx = 0;              // AudioRendered
    routing::NodeHandle handle;    // NodeMeta / NodeId
    routing::EffectMeta meta;      // NodeMeta
    routing::EffectId id;          // NodeId
};
class Receiver {
    friend class MpscClient;
    struct Chan {
        std::mutex m;
        std::condition_variable cv;
        std::deque<ClientMessage> q;
    };
    std::shared_ptr<Chan> ch_ = std::make_shared<Chan>();

public:
    ClientMessage recv() {   // blocks like Receiver::recv
        std::unique_lock<std::mutex> lk(ch_->m);
        ch_->cv.wait(lk, [&] { return !ch_->q.empty(); });
        ClientMessage msg = std::move(ch_->q.front());
        ch_->q.pop_front();
        return msg;
    }
    std::optional<ClientMessage> try_recv() {
        std::lock_guard<std::mutex> lk(ch_->m);
        if (ch_->q.empty()) return std::nullopt;
        ClientMessage msg = std::move(ch_->q.front());
        ch_->q.pop_front();
        return msg;
    }
};
class MpscClient : public Client {
    std::shared_ptr<Receiver::Chan> ch_;
    void send(ClientMessage msg) {
        {
            std::lock_guard<std::mutex> lk(ch_->m);
            ch_->q.push_back(std::move(msg));
        }
        ch_->cv.notify_one();
    }

public:
    static std::pair<MpscClient, Receiver> make() {
        Receiver rx;
        MpscClient c;
        c.ch_ = rx.ch_;
        return {std::move(c), std::move(rx)};
    }
    void audio_rendered(Array2 buffer, uint64_t idx) override {
        ClientMessage m{};
        m.kind = ClientMessage::AudioRendered;
        m.buffer = std::move(buffer);
        m.idx = idx;
        send(std::move(m));
    }
    void node_meta(const routing::NodeHandle &handle, const routing::EffectMeta &meta) override {
        ClientMessage m{};
        m.kind = ClientMessage::NodeMeta;
        m.handle = handle;
        m.meta = meta;
        send(std::move(m));
    }
    void node_id(const routing::NodeHandle &handle, const routing::EffectId &id) override {
        ClientMessage m{};
        m.kind = ClientMessage::NodeId;
        m.handle = handle;
        m.id = id;
        send(std::move(m));
    }
};

}  // namespace client

namespace dispatch {

using routing::Edge;
using routing::EffectId;
using routing::NodeHandle;

// dispatch.rs:45-63 (OSC address /routegraph/<...>)
struct OscRouteGraph {
    struct AddNode { NodeHandle handle; EffectId id; };      // "add_node"
    struct AddEdge { Edge edge; };                            // "add_edge"
    struct DelNode { NodeHandle handle; };                    // "del_node"
    struct DelEdge { Edge edge; };                            // "del_edge"
    struct QueryMeta { NodeHandle handle; };                  // "query_meta"
    struct QueryId { NodeHandle handle; };                    // "query_id"
    using Msg = std::variant<AddNode, AddEdge, DelNode, DelEdge, QueryMeta, QueryId>;
};
// dispatch.rs:65-77 (/renderer/render): range, number of output slots, inputs for slot 0..n
struct OscRenderer {
    struct RenderRange { uint64_t start, end; uint32_t num_slots; Jagged2 inputs; };
    using Msg = std::variant<RenderRange>;
};
// dispatch.rs:79-86 (/resman/add_dir)
struct OscResMan {
    struct AddDir { std::string dir; };
    using Msg = std::variant<AddDir>;
};
// dispatch.rs:30-43
using OscToplevel = std::variant<OscRouteGraph::Msg, OscRenderer::Msg, OscResMan::Msg>;

// The OSC address the reference's #[osc_address] attributes give each message (dispatch.rs:33-85).
inline std::string osc_address(const OscToplevel &msg) {
    if (auto *rg = std::get_if<OscRouteGraph::Msg>(&msg)) {
        static const char *names[] = {"add_node", "add_edge", "del_node", "del_edge", "query_meta", "query_id"};
        return std::string("/routegraph/") + names[rg->index()];
    }
    if (std::get_if<OscRenderer::Msg>(&msg)) return "/renderer/render";
    return "/resman/add_dir";
}

// dispatch.rs:89-93
struct Error : std::runtime_error {
    enum Kind { RouteGraphError, EffectError } kind;
    std::optional<routing::routegraph::ErrorKind> routegraph_kind;
    Error(Kind k, const std::string &m, std::optional<routing::routegraph::ErrorKind> rk = std::nullopt)
        : std::runtime_error(m), kind(k), routegraph_kind(rk) {}
};

// dispatch.rs:17-28,98-161,200-214
template <class R, class C>
class Dispatch {
    routing::RouteGraph routegraph_;
    R renderer_;
    resman::ResMan resman_;
    C client_;

public:
    Dispatch(R renderer, C client) : renderer_(std::move(renderer)), client_(std::move(client)) {}
    // "reference as written": AddEdge accepts edges that close a cycle (RouteGraph::CyclePolicy); default: WouldCycle
    void set_reference_as_written(bool on) {
        routegraph_.set_cycle_policy(on ? routing::RouteGraph::CyclePolicy::AsWritten : routing::RouteGraph::CyclePolicy::Documented);
    }
    R &renderer() { return renderer_; }
    C &client() { return client_; }
    resman::ResMan &resman() { return resman_; }

    // `msg.into()` of the reference (dispatch.rs:178-197): each concrete message wraps itself
    void dispatch(const OscRouteGraph::AddNode &m) { dispatch(OscToplevel(OscRouteGraph::Msg(m))); }
    void dispatch(const OscRouteGraph::AddEdge &m) { dispatch(OscToplevel(OscRouteGraph::Msg(m))); }
    void dispatch(const OscRouteGraph::DelNode &m) { dispatch(OscToplevel(OscRouteGraph::Msg(m))); }
    void dispatch(const OscRouteGraph::DelEdge &m) { dispatch(OscToplevel(OscRouteGraph::Msg(m))); }
    void dispatch(const OscRouteGraph::QueryMeta &m) { dispatch(OscToplevel(OscRouteGraph::Msg(m))); }
    void dispatch(const OscRouteGraph::QueryId &m) { dispatch(OscToplevel(OscRouteGraph::Msg(m))); }
    void dispatch(const OscRenderer::RenderRange &m) { dispatch(OscToplevel(OscRenderer::Msg(m))); }
    void dispatch(const OscResMan::AddDir &m) { dispatch(OscToplevel(OscResMan::Msg(m))); }

    // Process the message; throws dispatch::Error where the reference returns Err (dispatch.rs:111-161).
    void dispatch(const OscToplevel &msg) {
        try {
            if (auto *rg = std::get_if<OscRouteGraph::Msg>(&msg)) {
                if (auto *m = std::get_if<OscRouteGraph::AddNode>(rg)) {
                    routing::NodeData data = routing::Effect::from_id(m->id, resman_);
                    routegraph_.add_node(m->handle, data);
                    ptr(renderer_)->on_add_node(m->handle, data);
                } else if (auto *m = std::get_if<OscRouteGraph::AddEdge>(rg)) {
                    routegraph_.add_edge(m->edge);
                    ptr(renderer_)->on_add_edge(m->edge);
                } else if (auto *m = std::get_if<OscRouteGraph::DelNode>(rg)) {
                    routegraph_.del_node(m->handle);
                    ptr(renderer_)->on_del_node(m->handle);
                } else if (auto *m = std::get_if<OscRouteGraph::DelEdge>(rg)) {
                    routegraph_.del_edge(m->edge);
                    ptr(renderer_)->on_del_edge(m->edge);
                } else if (auto *m = std::get_if<OscRouteGraph::QueryMeta>(rg)) {
                    if (auto e = routegraph_.get_data(m->handle)) ptr(client_)->node_meta(m->handle, e->meta());
                } else if (auto *m = std::get_if<OscRouteGraph::QueryId>(rg)) {
                    if (auto e = routegraph_.get_data(m->handle)) ptr(client_)->node_id(m->handle, e->id());
                }
            } else if (auto *rm = std::get_if<OscRenderer::Msg>(&msg)) {
                const auto &m = std::get<OscRenderer::RenderRange>(*rm);
                Array2 buff = Array2::zeros(m.num_slots, (size_t)(m.end - m.start));
                ptr(renderer_)->fill_buffer(buff, m.start, m.inputs);
                ptr(client_)->audio_rendered(std::move(buff), m.start);
            } else if (auto *sm = std::get_if<OscResMan::Msg>(&msg)) {
                resman_.add_dir(std::get<OscResMan::AddDir>(*sm).dir);
            }
        } catch (const routing::routegraph::Error &e) {
            throw Error(Error::RouteGraphError, e.what(), e.kind);
        } catch (const routing::effect::NoMatchingEffect &e) {
            throw Error(Error::EffectError, e.what());
        }
    }

private:
    // R and C may be values or smart pointers
    template <class T> static T *ptr(T &v) { return &v; }
    template <class T> static T *ptr(std::unique_ptr<T> &v) { return v.get(); }
    template <class T> static T *ptr(std::shared_ptr<T> &v) { return v.get(); }
};

}  // namespace dispatch

using dispatch::Dispatch;
using client::Client;

}  // namespace friendship
