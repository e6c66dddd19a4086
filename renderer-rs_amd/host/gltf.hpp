// gltf.hpp -- C++ mirror of `Model::load` (crates/resources/src/model.rs:111-270): glTF 2.0 (.gltf + .bin or data: URIs)
// -> SoA meshes with the reference's defaults, materials (model.rs:273-309), AABB.  Node transforms are NOT applied
// (model.rs:135-144).  Images: the reference lets gltf::import decode them and discards the result (model.rs:120), and so
// does load(path); load(path, ImagePolicy::Decode) keeps them -- every images[i] (file URI, data: URI, bufferView) goes
// through image_decode.hpp to RGBA8 and each material records the image behind its five texture slots
// (model_pbr.hlsl:62-95, t0..t4).  A file that is absent becomes an empty optional and is listed in missing_images.
// Header-only, no dependency: the JSON reader below covers the subset glTF uses (objects, arrays, strings, numbers,
// true/false/null).
// the caller side of the hot path -- its output feeds Mesh::interleave() -> `Vertex` streams.
#ifndef MIRHI_GLTF_HPP
#define MIRHI_GLTF_HPP

#include <cfloat>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "image_decode.hpp"
#include "mirhi.hpp"

namespace mirhi {
namespace resources {

struct ResourceError : std::runtime_error { using std::runtime_error::runtime_error; };   // crates/resources/src/error.rs:6-40

namespace json {
struct Value;
using Ptr = std::shared_ptr<Value>;
struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false; double num = 0; std::string str; std::vector<Ptr> arr; std::map<std::string, Ptr> obj;
    bool has(const std::string& k) const { return kind == Object && obj.count(k); }
    const Value& at(const std::string& k) const {
        auto it = obj.find(k);
        if (kind != Object || it == obj.end()) throw ResourceError("glTF: missing key '" + k + "'");
        return *it->second;
    }
    const Value& at(size_t i) const { if (kind != Array || i >= arr.size()) throw ResourceError("glTF: index out of range"); return *arr[i]; }
    size_t size() const { return kind == Array ? arr.size() : 0; }
    double number(double dflt) const { return kind == Number ? num : dflt; }
    double get(const std::string& k, double dflt) const { return has(k) ? at(k).number(dflt) : dflt; }
};
class Parser {
public:
    explicit Parser(const std::string& s) : s_(s) {}
    Ptr parse() { Ptr v = value(); ws(); if (i_ != s_.size()) fail("trailing characters"); return v; }
private:
    const std::string& s_; size_t i_ = 0;
    [[noreturn]] void fail(const char* what) const { throw ResourceError(std::string("glTF JSON: ") + what + " at byte " + std::to_string(i_)); }
    void ws() { while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) i_++; }
    bool eat(char c) { ws(); if (i_ < s_.size() && s_[i_] == c) { i_++; return true; } return false; }
    Ptr value() {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        auto v = std::make_shared<Value>();
        const char c = s_[i_];
        if (c == '{') {
            i_++; v->kind = Value::Object;
            if (eat('}')) return v;
            do { ws(); std::string k = string(); if (!eat(':')) fail("expected ':'"); v->obj[k] = value(); } while (eat(','));
            if (!eat('}')) fail("expected '}'");
        } else if (c == '[') {
            i_++; v->kind = Value::Array;
            if (eat(']')) return v;
            do { v->arr.push_back(value()); } while (eat(','));
            if (!eat(']')) fail("expected ']'");
        } else if (c == '"') { v->kind = Value::String; v->str = string(); }
        else if (s_.compare(i_, 4, "true") == 0) { v->kind = Value::Bool; v->b = true; i_ += 4; }
        else if (s_.compare(i_, 5, "false") == 0) { v->kind = Value::Bool; i_ += 5; }
        else if (s_.compare(i_, 4, "null") == 0) { i_ += 4; }
        else {
            size_t used = 0;
            try { v->num = std::stod(s_.substr(i_, 64), &used); } catch (...) { fail("bad number"); }
            v->kind = Value::Number; i_ += used;
        }
        return v;
    }
    std::string string() {
        if (i_ >= s_.size() || s_[i_] != '"') fail("expected string");
        i_++;
        std::string out;
        while (i_ < s_.size() && s_[i_] != '"') {
            char c = s_[i_++];
            if (c == '\\') {
                if (i_ >= s_.size()) fail("bad escape");
                const char e = s_[i_++];
                switch (e) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': { if (i_ + 4 > s_.size()) fail("bad \\u"); unsigned cp = std::stoul(s_.substr(i_, 4), nullptr, 16); i_ += 4;
                                if (cp < 0x80) out += (char)cp; else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                                else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); } break; }
                    default: out += e;
                }
            } else out += c;
        }
        if (i_ >= s_.size()) fail("unterminated string");
        i_++;
        return out;
    }
};
}  // namespace json

struct LoadedMesh : Mesh { std::optional<size_t> material_index; size_t vertex_count() const { return positions.size(); } size_t triangle_count() const { return indices.size() / 3; } };

enum class ImagePolicy { Discard, Decode };

// what a glTF material says about textures and alpha, next to the reference's `Material` (factors only, material.rs:6-30)
struct MaterialTextures {
    std::optional<size_t> base_color, metallic_roughness, normal, occlusion, emissive;   // indices into Model::images
    float normal_scale = 1.0f, occlusion_strength = 1.0f, alpha_cutoff = 0.5f;
    std::string alpha_mode = "OPAQUE";
    bool double_sided = false;
};

struct Model {   // model.rs:46-109
    std::vector<LoadedMesh> meshes;
    std::vector<Material> materials;
    std::vector<MaterialTextures> material_textures;      // parallel to materials; filled by ImagePolicy::Decode
    std::vector<std::optional<ImageData>> images;         // one per glTF image (ImagePolicy::Decode)
    std::vector<std::string> missing_images;
    Vec3 aabb_min{FLT_MAX, FLT_MAX, FLT_MAX}, aabb_max{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    size_t total_vertices() const { size_t n = 0; for (auto& m : meshes) n += m.vertex_count(); return n; }
    size_t total_triangles() const { size_t n = 0; for (auto& m : meshes) n += m.triangle_count(); return n; }

    // Every failure -- missing file, malformed JSON, an index or an accessor that points outside its array or buffer, an
    // allocation a hostile count provokes -- leaves as ResourceError, like `gltf::import(path)?` does in the reference.
    static Model load(const std::string& path, ImagePolicy policy = ImagePolicy::Discard) {
        try { return load_unchecked(path, policy); }
        catch (const ResourceError&) { throw; }
        catch (const std::exception& e) { throw ResourceError("Failed to load glTF " + path + ": " + e.what()); }
    }

private:
    static Model load_unchecked(const std::string& path, ImagePolicy policy) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw ResourceError("File not found: " + path);                         // model.rs:113-115
        std::stringstream ss; ss << f.rdbuf();
        const std::string text = ss.str();
        const json::Ptr docp = json::Parser(text).parse();
        const json::Value& doc = *docp;
        const std::string base = path.find_last_of("/\\") == std::string::npos ? std::string(".") : path.substr(0, path.find_last_of("/\\"));
        std::vector<std::vector<uint8_t>> buffers;
        if (doc.has("buffers")) for (size_t i = 0; i < doc.at("buffers").size(); i++) {
            const json::Value& b = doc.at("buffers").at(i);
            const std::string uri = b.has("uri") ? b.at("uri").str : std::string();
            if (uri.rfind("data:", 0) == 0) buffers.push_back(base64(uri.substr(uri.find(',') + 1)));
            else {
                std::ifstream bf(base + "/" + uri, std::ios::binary);
                if (!bf) throw ResourceError("Failed to load glTF " + path + ": buffer '" + uri + "' not found");
                buffers.emplace_back((std::istreambuf_iterator<char>(bf)), std::istreambuf_iterator<char>());
            }
        }
        Model model;
        if (doc.has("materials")) for (size_t i = 0; i < doc.at("materials").size(); i++) {      // model.rs:273-309
            const json::Value& m = doc.at("materials").at(i);
            Material out;
            out.metallic = 1.0f; out.roughness = 1.0f;                                            // glTF defaults
            if (m.has("pbrMetallicRoughness")) {
                const json::Value& p = m.at("pbrMetallicRoughness");
                if (p.has("baseColorFactor")) { const json::Value& c = p.at("baseColorFactor"); out.base_color = {(float)c.at(0).num, (float)c.at(1).num, (float)c.at(2).num, (float)c.at(3).num}; }
                out.metallic = (float)p.get("metallicFactor", 1.0); out.roughness = (float)p.get("roughnessFactor", 1.0);
            }
            out.ao = 1.0f;
            if (m.has("emissiveFactor")) { const json::Value& e = m.at("emissiveFactor"); out.emissive = {(float)e.at(0).num, (float)e.at(1).num, (float)e.at(2).num, 1.0f}; }
            else out.emissive = {0, 0, 0, 1.0f};
            model.materials.push_back(out);
            if (policy == ImagePolicy::Decode) {
                MaterialTextures t;
                auto source = [&](const json::Value& owner, const char* key) -> std::optional<size_t> {
                    if (!owner.has(key) || !owner.at(key).has("index") || !doc.has("textures")) return std::nullopt;
                    const json::Value& tex = doc.at("textures").at((size_t)owner.at(key).at("index").num);
                    if (!tex.has("source")) return std::nullopt;
                    return (size_t)tex.at("source").num;
                };
                if (m.has("pbrMetallicRoughness")) {
                    t.base_color = source(m.at("pbrMetallicRoughness"), "baseColorTexture");
                    t.metallic_roughness = source(m.at("pbrMetallicRoughness"), "metallicRoughnessTexture");
                }
                t.normal = source(m, "normalTexture"); t.occlusion = source(m, "occlusionTexture"); t.emissive = source(m, "emissiveTexture");
                if (m.has("normalTexture")) t.normal_scale = (float)m.at("normalTexture").get("scale", 1.0);
                if (m.has("occlusionTexture")) t.occlusion_strength = (float)m.at("occlusionTexture").get("strength", 1.0);
                if (m.has("alphaMode")) t.alpha_mode = m.at("alphaMode").str;
                t.alpha_cutoff = (float)m.get("alphaCutoff", 0.5);
                t.double_sided = m.has("doubleSided") && m.at("doubleSided").b;
                model.material_textures.push_back(t);
            }
        }
        if (policy == ImagePolicy::Decode && doc.has("images")) for (size_t i = 0; i < doc.at("images").size(); i++) {
            const json::Value& img = doc.at("images").at(i);
            std::vector<uint8_t> bytes; std::string label;
            if (img.has("uri")) {
                const std::string uri = img.at("uri").str; label = uri;
                if (uri.rfind("data:", 0) == 0) { bytes = base64(uri.substr(uri.find(',') + 1)); label = "data: URI"; }
                else {
                    std::ifstream imf(base + "/" + percent_decode(uri), std::ios::binary);
                    if (!imf) { model.images.emplace_back(std::nullopt); model.missing_images.push_back(uri); continue; }
                    bytes.assign((std::istreambuf_iterator<char>(imf)), std::istreambuf_iterator<char>());
                }
            } else if (img.has("bufferView")) {
                const json::Value& bv = doc.at("bufferViews").at((size_t)img.at("bufferView").num);
                const std::vector<uint8_t>& raw = buffers.at((size_t)bv.at("buffer").num);
                const size_t start = (size_t)bv.get("byteOffset", 0), len = (size_t)bv.at("byteLength").num;
                if (start + len > raw.size()) throw ResourceError("glTF: image bufferView exceeds its buffer");
                bytes.assign(raw.begin() + start, raw.begin() + start + len); label = "bufferView";
            } else throw ResourceError("glTF: image " + std::to_string(i) + " has neither uri nor bufferView");
            try { model.images.emplace_back(decode_image(bytes.data(), bytes.size())); }
            catch (const ImageError& e) { throw ResourceError("Failed to decode image " + label + " of " + path + ": " + e.what()); }
        }
        if (doc.has("meshes")) for (size_t mi = 0; mi < doc.at("meshes").size(); mi++) {
            const json::Value& mesh = doc.at("meshes").at(mi);
            if (!mesh.has("primitives")) continue;
            for (size_t pi = 0; pi < mesh.at("primitives").size(); pi++) {
                const json::Value& prim = mesh.at("primitives").at(pi);
                const json::Value* attrs = prim.has("attributes") ? &prim.at("attributes") : nullptr;
                if (!attrs || !attrs->has("POSITION")) throw ResourceError("No position data");      // model.rs:150-153
                LoadedMesh out;
                const std::vector<float> pos = floats(doc, buffers, (size_t)attrs->at("POSITION").num, 3);
                const size_t n = pos.size() / 3;
                if (n == 0) continue;
                out.positions.resize(n);
                for (size_t i = 0; i < n; i++) {
                    out.positions[i] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
                    model.aabb_min = {std::min(model.aabb_min.x, pos[3 * i]), std::min(model.aabb_min.y, pos[3 * i + 1]), std::min(model.aabb_min.z, pos[3 * i + 2])};
                    model.aabb_max = {std::max(model.aabb_max.x, pos[3 * i]), std::max(model.aabb_max.y, pos[3 * i + 1]), std::max(model.aabb_max.z, pos[3 * i + 2])};
                }
                out.normals.assign(n, Vec3{0, 1, 0}); out.tex_coords.assign(n, Vec2{}); out.tangents.assign(n, Vec4{1, 0, 0, 1});   // model.rs:160-215
                if (attrs->has("NORMAL")) { const auto v = floats(doc, buffers, (size_t)attrs->at("NORMAL").num, 3); for (size_t i = 0; i < n && 3 * i + 2 < v.size(); i++) out.normals[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]}; }
                if (attrs->has("TEXCOORD_0")) { const auto v = floats(doc, buffers, (size_t)attrs->at("TEXCOORD_0").num, 2); for (size_t i = 0; i < n && 2 * i + 1 < v.size(); i++) out.tex_coords[i] = {v[2 * i], v[2 * i + 1]}; }
                if (attrs->has("TANGENT")) { const auto v = floats(doc, buffers, (size_t)attrs->at("TANGENT").num, 4); for (size_t i = 0; i < n && 4 * i + 3 < v.size(); i++) out.tangents[i] = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]}; }
                if (prim.has("indices")) {
                    out.indices = indices(doc, buffers, (size_t)prim.at("indices").num);
                    // an index beyond the primitive's vertices would make the GPU's vertex fetch read out of bounds (a device fault,
                    // not an error code): refuse the asset here
                    for (const uint32_t ix : out.indices)
                        if (ix >= n) throw ResourceError("Index " + std::to_string(ix) + " out of range for a primitive with " + std::to_string(n) + " vertices");
                }
                else { out.indices.resize(n); for (size_t i = 0; i < n; i++) out.indices[i] = (uint32_t)i; }
                if (prim.has("material")) out.material_index = (size_t)prim.at("material").num;
                model.meshes.push_back(std::move(out));
            }
        }
        if (model.meshes.empty()) throw ResourceError("No meshes found in glTF file");              // model.rs:262-264
        return model;
    }

private:
    static std::string percent_decode(const std::string& s) {
        std::string out;
        for (size_t i = 0; i < s.size(); i++) {
            if (s[i] == '%' && i + 2 < s.size() + 0 && isxdigit((unsigned char)s[i + 1]) && isxdigit((unsigned char)s[i + 2])) {
                out.push_back((char)std::stoi(s.substr(i + 1, 2), nullptr, 16)); i += 2;
            } else out.push_back(s[i]);
        }
        return out;
    }
    static std::vector<uint8_t> base64(const std::string& s) {
        std::vector<uint8_t> out; uint32_t acc = 0; int bits = 0;
        for (char c : s) {
            int v = c >= 'A' && c <= 'Z' ? c - 'A' : c >= 'a' && c <= 'z' ? c - 'a' + 26 : c >= '0' && c <= '9' ? c - '0' + 52 : c == '+' ? 62 : c == '/' ? 63 : -1;
            if (v < 0) continue;
            acc = (acc << 6) | (uint32_t)v; bits += 6;
            if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
        }
        return out;
    }
    struct View { const uint8_t* p; size_t count, stride, comps, csize; int ctype; };
    static View view(const json::Value& doc, const std::vector<std::vector<uint8_t>>& buffers, size_t index) {
        const json::Value& acc = doc.at("accessors").at(index);
        const int ctype = (int)acc.at("componentType").num;
        const std::string& type = acc.at("type").str;
        const size_t comps = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT2" ? 4 : type == "MAT3" ? 9 : 16;
        const size_t csize = (ctype == 5120 || ctype == 5121) ? 1 : (ctype == 5122 || ctype == 5123) ? 2 : 4;
        const size_t count = (size_t)acc.at("count").num;
        if (count > (size_t(1) << 28)) throw ResourceError("glTF: accessor count out of range");
        if (!acc.has("bufferView")) return {nullptr, count, comps * csize, comps, csize, ctype};
        const json::Value& bv = doc.at("bufferViews").at((size_t)acc.at("bufferView").num);
        const std::vector<uint8_t>& raw = buffers.at((size_t)bv.at("buffer").num);
        const size_t start = (size_t)bv.get("byteOffset", 0) + (size_t)acc.get("byteOffset", 0);
        size_t stride = (size_t)bv.get("byteStride", 0);
        if (!stride) stride = comps * csize;
        // (overflow-safe: every element takes at least one byte of the buffer, so a count beyond its size is wrong at once)
        if (count > raw.size() || start > raw.size() || stride > raw.size() ||
            (count && (count - 1) > (raw.size() - start) / (stride ? stride : 1)) ||
            (count && start + stride * (count - 1) + comps * csize > raw.size())) throw ResourceError("glTF: accessor exceeds its buffer");
        return {raw.data() + start, count, stride, comps, csize, ctype};
    }
    static std::vector<float> floats(const json::Value& doc, const std::vector<std::vector<uint8_t>>& buffers, size_t index, size_t want) {
        const View v = view(doc, buffers, index);
        std::vector<float> out(v.count * want, 0.0f);
        if (!v.p) return out;
        for (size_t i = 0; i < v.count; i++)
            for (size_t c = 0; c < want && c < v.comps; c++) {
                const uint8_t* e = v.p + i * v.stride + c * v.csize;
                float x;
                switch (v.ctype) {                       // plain numeric conversion, as `.astype(f32)` / gltf's reader for floats
                    case 5126: memcpy(&x, e, 4); break;
                    case 5125: { uint32_t u; memcpy(&u, e, 4); x = (float)u; break; }
                    case 5123: { uint16_t u; memcpy(&u, e, 2); x = (float)u; break; }
                    case 5122: { int16_t u; memcpy(&u, e, 2); x = (float)u; break; }
                    case 5121: x = (float)*e; break;
                    default: x = (float)*(const int8_t*)e;
                }
                out[i * want + c] = x;
            }
        return out;
    }
    static std::vector<uint32_t> indices(const json::Value& doc, const std::vector<std::vector<uint8_t>>& buffers, size_t index) {
        const View v = view(doc, buffers, index);
        std::vector<uint32_t> out(v.count, 0u);
        if (!v.p) return out;
        for (size_t i = 0; i < v.count; i++) {
            const uint8_t* e = v.p + i * v.stride;
            if (v.csize == 4) memcpy(&out[i], e, 4);
            else if (v.csize == 2) { uint16_t u; memcpy(&u, e, 2); out[i] = u; }
            else out[i] = *e;
        }
        return out;
    }
};

}  // namespace resources
}  // namespace mirhi
#endif
