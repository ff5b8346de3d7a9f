// scene_io.cpp — scene/asset front-end of the host mirror; see scene_io.h for the reference map.
#include "scene_io.h"
#include "../fpt_math.h"
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <tuple>
#include <unordered_map>

namespace fermat {

namespace {

// ---- a scanner with the semantics of the reference's fscanf("%s") / fscanf("%f") / fgets() mix (src/mesh/MeshBase.cpp) ----------
struct Scanner
{
	std::string s; size_t p = 0;
	bool open(const std::string& filename)
	{
		FILE* f = std::fopen(filename.c_str(), "rb");
		if (!f) return false;
		std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
		s.resize(n > 0 ? size_t(n) : 0);
		if (n > 0 && std::fread(&s[0], 1, size_t(n), f) != size_t(n)) { std::fclose(f); return false; }
		std::fclose(f);
		return true;
	}
	static bool space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
	void skip_ws() { while (p < s.size() && space(s[p])) ++p; }
	bool token(std::string& t)                           // fscanf(file, "%s", buf)
	{
		skip_ws();
		if (p >= s.size()) return false;
		const size_t b = p;
		while (p < s.size() && !space(s[p])) ++p;
		t.assign(s, b, p - b);
		return true;
	}
	std::string line()                                   // fgets(): the rest of the current line, newline consumed
	{
		const size_t b = p;
		while (p < s.size() && s[p] != '\n') ++p;
		std::string r(s, b, p - b);
		if (p < s.size()) ++p;
		return r;
	}
	bool read_float(float& v)                            // fscanf(file, "%f", &v): consumes only what parses
	{
		skip_ws();
		if (p >= s.size()) return false;
		const char* b = s.c_str() + p; char* e = nullptr;
		const float r = std::strtof(b, &e);
		if (e == b) return false;
		v = r; p += size_t(e - b);
		return true;
	}
	bool read_int(int& v)
	{
		skip_ws();
		if (p >= s.size()) return false;
		const char* b = s.c_str() + p; char* e = nullptr;
		const long r = std::strtol(b, &e, 10);
		if (e == b) return false;
		v = int(r); p += size_t(e - b);
		return true;
	}
	// does the input continue with an integer (what "while (fscanf(file, "%d...") > 0)" tests)?
	bool next_is_int()
	{
		skip_ws();
		if (p >= s.size()) return false;
		size_t q = p;
		if (s[q] == '-' || s[q] == '+') ++q;
		return q < s.size() && s[q] >= '0' && s[q] <= '9';
	}
};

// sscanf(buf, "%s %s", buf2, buf2) on the fgets()'d remainder of a keyword line: the LAST of the first two words wins
std::string second_word(const std::string& line)
{
	std::istringstream ss(line);
	std::string a, b;
	ss >> a;
	if (ss >> b) return b;
	return a;
}
std::string first_word(const std::string& line) { std::istringstream ss(line); std::string a; ss >> a; return a; }

std::string directory_of(const std::string& filename)       // directoryOfFilePath: up to and including the last separator
{
	const size_t k = filename.find_last_of("/\\");
	return k == std::string::npos ? std::string() : filename.substr(0, k + 1);
}

// ---- MTL : MeshBase::loadMaterials (src/mesh/MeshBase.cpp:492-713) --------------------------------------------------------------
// staging[0] is a default-valued material named "<default name>_0" that the reference also appends to the material table.
bool read_mtl(const std::string& filename, std::vector<MeshMaterialParams>& staging)
{
	Scanner sc;
	if (!sc.open(filename)) return false;
	unsigned num_materials = 1;
	{
		Scanner cnt = sc; std::string t;
		while (cnt.token(t)) { if (t[0] == 'n') ++num_materials; cnt.line(); }
	}
	staging.assign(num_materials, MeshMaterialParams());
	for (size_t i = 0; i < staging.size(); ++i) staging[i].name += "_" + std::to_string(i);
	int cur = 0;
	std::string t;
	auto rd3 = [&](float* d) { sc.read_float(d[0]); sc.read_float(d[1]); sc.read_float(d[2]); };
	while (sc.token(t))
	{
		MeshMaterialParams& m = staging[size_t(cur)];
		switch (t[0])
		{
		case '#': sc.line(); break;
		case 'n': { const std::string name = second_word(sc.line()); if (size_t(cur + 1) < staging.size()) ++cur; staging[size_t(cur)].name = name; break; }
		case 'N':
			if (t.size() > 1 && t[1] == 's') sc.read_float(m.phong_exponent);
			else if (t.size() > 1 && t[1] == 'i') sc.read_float(m.index_of_refraction);
			break;
		case 'T':
			if (t.size() > 1 && t[1] == 'r') { float tr = 0.0f; sc.read_float(tr); m.opacity = 1.0f - tr; }
			else if (t.size() > 1 && t[1] == 'd') rd3(m.diffuse_trans);
			break;
		case 'd': sc.read_float(m.opacity); break;
		case 'i': sc.read_int(m.shading_type); break;
		case 'r': { float r = 0.0f; sc.read_float(r); m.reflectivity[0] = m.reflectivity[1] = m.reflectivity[2] = r; break; }
		case 'e': rd3(m.emissive); break;
		case 'f': sc.read_int(m.flags); break;
		case 'm':
		{
			MeshTextureMap* map = nullptr;
			if (t == "map_Ka") map = &m.ambient_map;
			else if (t == "map_Kd") map = &m.diffuse_map;
			else if (t == "map_Ks") map = &m.specular_map;
			else if (t == "map_Ke") map = &m.emissive_map;
			else if (t == "map_Td") map = &m.diffuse_trans_map;
			else if (t == "map_D" || t == "map_d") map = &m.opacity_map;
			else if (t == "map_Bump" || t == "map_bump") map = &m.bump_map;
			else { std::fprintf(stderr, "Unknown map: \"%s\"\n", t.c_str()); break; }
			std::string name;
			sc.token(name);
			if (name == "-s") { sc.read_float(map->scaling[0]); sc.read_float(map->scaling[1]); sc.token(name); }
			map->name = name;
			break;
		}
		case 'K':
			switch (t.size() > 1 ? t[1] : '\0')
			{
			case 'd': rd3(m.diffuse); break;
			case 's': rd3(m.specular); break;
			case 'a': rd3(m.ambient); break;
			case 'e': rd3(m.emissive); break;
			case 'r': rd3(m.reflectivity); break;
			default: sc.line(); break;
			}
			break;
		default: sc.line(); break;
		}
	}
	return true;
}

// insert_texture (src/mesh/MeshStorage.cpp:46-81): first-use order, one slot per distinct file name
fpt_texture_ref insert_texture(MeshStorage& mesh, const MeshTextureMap& tex)
{
	fpt_texture_ref r; r.texture = 0xFFFFFFFFu; r._pad = 0; r.scaling[0] = tex.scaling[0]; r.scaling[1] = tex.scaling[1];
	if (tex.name.empty()) return r;
	auto it = mesh.textures_map.find(tex.name);
	if (it == mesh.textures_map.end())
	{
		const uint32 id = uint32(mesh.textures.size());
		mesh.textures_map.insert(std::make_pair(tex.name, id));
		mesh.textures.push_back(tex.name);
		r.texture = id;
	}
	else r.texture = it->second;
	return r;
}

fpt_material make_material(MeshStorage& mesh, const MeshMaterialParams& p, bool keep_bump)
{
	fpt_material m; std::memset(&m, 0, sizeof(m));
	for (int c = 0; c < 3; ++c)
	{
		m.ambient[c] = p.ambient[c]; m.diffuse[c] = p.diffuse[c]; m.diffuse_trans[c] = p.diffuse_trans[c];
		m.specular[c] = p.specular[c]; m.emissive[c] = p.emissive[c]; m.reflectivity[c] = p.reflectivity[c];
	}
	m.roughness = p.phong_exponent ? 1.0f / p.phong_exponent : 1.0f;       // 1/powf(Ns,1), src/mesh/MeshStorage.cpp:163
	m.index_of_refraction = p.index_of_refraction; m.opacity = p.opacity; m.flags = p.flags;
	m.ambient_map = insert_texture(mesh, p.ambient_map);
	m.diffuse_map = insert_texture(mesh, p.diffuse_map);
	m.diffuse_trans_map = insert_texture(mesh, p.diffuse_trans_map);
	m.specular_map = insert_texture(mesh, p.specular_map);
	m.emissive_map = insert_texture(mesh, p.emissive_map);
	if (keep_bump) m.bump_map = insert_texture(mesh, p.bump_map);
	else { m.bump_map.texture = 0xFFFFFFFFu; m.bump_map._pad = 0; m.bump_map.scaling[0] = m.bump_map.scaling[1] = 1.0f; }
	return m;
}

struct Corner { int v, t, n; };

} // namespace

// ---- OBJ : MeshBase::loadInfoFromObj / loadDataFromObj (src/mesh/MeshBase.cpp:728-1400) + MeshLoader::allocateData ---------------
// Triangles are stored group by group; a group is "<g name>:<usemtl name>" (kKeepGroups) and groups live in a std::map, i.e. in
// lexicographic name order — NOT in file order.  Faces are fanned.  Material 0 is the inserted default material, 1 the MTL
// staging default, then the library's materials in file order.
static void load_obj(const std::string& filename, MeshStorage& mesh)
{
	Scanner sc;
	if (!sc.open(filename)) throw MeshException("unable to open file: " + filename);
	mesh = MeshStorage();

	std::vector<MeshMaterialParams> params(1);                 // the default material (insertDefaultMaterial = true)
	std::map<std::string, int> material_numbers;
	material_numbers[params[0].name] = 0;
	int material_count = 1;

	struct Group { std::vector<int> v, n, t, m; };
	std::map<std::string, Group> groups;
	std::vector<float> P, N, T;
	bool any_normal_index = false, any_tex_index = false;

	std::string group_base = "null-group", material_name = params[0].name;
	int material_number = 0;
	Group* cur = &groups[group_base];
	std::string tok;
	while (sc.token(tok))
	{
		switch (tok[0])
		{
		case '#': sc.line(); break;
		case 'v':
			if (tok.size() == 1) { float f[3] = { 0, 0, 0 }; sc.read_float(f[0]); sc.read_float(f[1]); sc.read_float(f[2]); P.insert(P.end(), f, f + 3); }
			else if (tok[1] == 'n') { float f[3] = { 0, 0, 0 }; sc.read_float(f[0]); sc.read_float(f[1]); sc.read_float(f[2]); N.insert(N.end(), f, f + 3); }
			else if (tok[1] == 't') { float f[2] = { 0, 0 }; sc.read_float(f[0]); sc.read_float(f[1]); T.insert(T.end(), f, f + 2); }
			break;
		case 'm':
		{
			const std::string lib = second_word(sc.line());
			std::vector<MeshMaterialParams> staging;
			if (read_mtl(directory_of(filename) + lib, staging))
				for (const MeshMaterialParams& s : staging)
				{
					material_numbers.insert(std::make_pair(s.name, int(params.size())));
					params.push_back(s);
				}
			break;
		}
		case 'u':
		{
			material_name = second_word(sc.line());
			auto it = material_numbers.find(material_name);
			if (it == material_numbers.end()) it = material_numbers.insert(std::make_pair(material_name, material_count++)).first;
			material_number = it->second;
			cur = &groups[group_base + ":" + material_name];
			break;
		}
		case 'o': sc.line(); break;
		case 'g':
			group_base = first_word(sc.line());
			cur = &groups[group_base + ":" + material_name];
			break;
		case 'f':
		{
			std::vector<Corner> corners;
			std::string c;
			const int nv = int(P.size() / 3), nn = int(N.size() / 3), nt = int(T.size() / 2);
			auto fix = [](int i, int count) { return i >= 0 ? i - 1 : count + i; };
			int mode = 0;                                          // 1: v//n  2: v/t/n  3: v/t  4: v
			bool first = true;
			for (;;)
			{
				if (first) { if (!sc.token(c)) break; }
				else { if (!sc.next_is_int() || !sc.token(c)) break; }
				int v = 0, t = 0, n = 0;
				if (first)
				{
					if (c.find("//") != std::string::npos) mode = 1;
					else if (std::sscanf(c.c_str(), "%d/%d/%d", &v, &t, &n) == 3) mode = 2;
					else if (std::sscanf(c.c_str(), "%d/%d", &v, &t) == 2) mode = 3;
					else mode = 4;
					first = false;
				}
				v = t = n = 0;
				Corner k; k.v = k.t = k.n = -1;
				if (mode == 1) { std::sscanf(c.c_str(), "%d//%d", &v, &n); k.v = fix(v, nv); k.n = fix(n, nn); }
				else if (mode == 2) { std::sscanf(c.c_str(), "%d/%d/%d", &v, &t, &n); k.v = fix(v, nv); k.t = fix(t, nt); k.n = fix(n, nn); }
				else if (mode == 3) { std::sscanf(c.c_str(), "%d/%d", &v, &t); k.v = fix(v, nv); k.t = fix(t, nt); }
				else { std::sscanf(c.c_str(), "%d", &v); k.v = fix(v, nv); }
				corners.push_back(k);
			}
			if (mode == 1 || mode == 2) any_normal_index = true;
			if (mode == 2 || mode == 3) any_tex_index = true;
			for (size_t i = 1; i + 1 < corners.size(); ++i)            // triangle fan: (first, previous last, new)
			{
				const Corner& a = corners[0]; const Corner& b = corners[i]; const Corner& d = corners[i + 1];
				cur->v.insert(cur->v.end(), { a.v, b.v, d.v, 0 });
				cur->n.insert(cur->n.end(), { a.n, b.n, d.n, 0 });
				cur->t.insert(cur->t.end(), { a.t, b.t, d.t, 0 });
				cur->m.push_back(material_number);
			}
			break;
		}
		default: sc.line(); break;
		}
	}
	(void)any_normal_index; (void)any_tex_index;

	mesh.num_vertices = int(P.size() / 3);
	mesh.num_normals = int(N.size() / 3);
	mesh.num_texture_coordinates = int(T.size() / 2);
	mesh.vertex_data.assign(size_t(mesh.num_vertices) * 4, 0.0f);
	for (int i = 0; i < mesh.num_vertices; ++i) for (int c = 0; c < 3; ++c) mesh.vertex_data[size_t(i) * 4 + c] = P[size_t(i) * 3 + c];
	mesh.normal_data = N;
	mesh.texture_data = T;
	mesh.group_offsets.clear();
	// Face indices are range-checked against the final element counts before anything indexes with them (the reference trusts its
	// input here and reads out of bounds on a malformed or truncated OBJ; compress_tex / unify_vertex_attributes below index
	// vertex_data, normal_data and texture_data with these values).  Position indices must exist; -1 marks an absent normal / texcoord.
	for (auto& g : groups)
	{
		for (size_t i = 0; i < g.second.v.size(); ++i)
		{
			if ((i & 3) == 3) continue;
			const int v = g.second.v[i], n = g.second.n[i], t = g.second.t[i];
			if (v < 0 || v >= mesh.num_vertices || n < -1 || n >= mesh.num_normals || t < -1 || t >= mesh.num_texture_coordinates)
				throw MeshException("face index out of range in OBJ file (" + filename + ")");
		}
	}
	for (auto& g : groups)                                          // std::map order == the reference's group order
	{
		if (g.second.m.empty()) continue;                           // PruneEmptyGroupsFunctor
		mesh.group_names.push_back(g.first);
		mesh.group_offsets.push_back(mesh.num_triangles);
		mesh.vertex_indices.insert(mesh.vertex_indices.end(), g.second.v.begin(), g.second.v.end());
		mesh.normal_indices.insert(mesh.normal_indices.end(), g.second.n.begin(), g.second.n.end());
		mesh.texture_indices.insert(mesh.texture_indices.end(), g.second.t.begin(), g.second.t.end());
		mesh.material_indices.insert(mesh.material_indices.end(), g.second.m.begin(), g.second.m.end());
		mesh.num_triangles += int(g.second.m.size());
	}
	mesh.group_offsets.push_back(mesh.num_triangles);
	// without normals / texture coordinates the reference keeps no index stream at all
	if (mesh.num_normals == 0) mesh.normal_indices.clear();
	if (mesh.num_texture_coordinates == 0) mesh.texture_indices.clear();

	// a usemtl that names no library material still owns a material slot (default values)
	if (int(params.size()) < material_count) params.resize(size_t(material_count));
	for (const MeshMaterialParams& p : params)
	{
		mesh.materials.push_back(make_material(mesh, p, true));
		mesh.material_names.push_back(p.name);
	}
}

// ---- PLY : MeshBase::loadFromPly (src/mesh/MeshBase.cpp:191-340,1416-1520) over rply 1.01 (src/mesh/rply-1.01/rply.c) -----------
// rply's reading rules restated: magic "ply\n"; header words split at " \n\r\t"; `format <ascii|binary_little_endian|
// binary_big_endian> 1.0`; `element <name> <count>` followed by `property <type> <name>` / `property list <len type> <value type>
// <name>` lines, `comment` / `obj_info` lines anywhere; after the word `end_header` exactly ONE delimiter byte is consumed and the
// data starts.  Every value goes through a double.  The loader's callbacks are a small state machine, reproduced as such: x, y
// write the current vertex and z advances it (likewise nx ny nz and s t / u v), the first three entries of every `vertex_indices`
// list write the current triangle and the third advances it (longer faces lose their tail, shorter ones are overwritten);
// normal and texture-coordinate indices are the vertex indices.  One group "null-group", one (default) material.
namespace {

enum PlyType { P_INT8, P_UINT8, P_INT16, P_UINT16, P_INT32, P_UINT32, P_FLOAT32, P_FLOAT64, P_LIST, P_NONE };

PlyType ply_type(const std::string& w)
{
	static const char* const names[] = { "int8", "uint8", "int16", "uint16", "int32", "uint32", "float32", "float64",
	                                     "char", "uchar", "short", "ushort", "int", "uint", "float", "double" };
	for (int i = 0; i < 16; ++i) if (w == names[i]) return PlyType(i & 7);
	return w == "list" ? P_LIST : P_NONE;
}

struct PlyProperty { std::string name; PlyType type = P_NONE, length_type = P_NONE, value_type = P_NONE; };
struct PlyElement { std::string name; long count = 0; std::vector<PlyProperty> props; };

struct PlyReader
{
	std::vector<unsigned char> buf;
	size_t pos = 0;
	int mode = 0;                   // 0 ascii, 1 little endian, 2 big endian
	std::vector<PlyElement> elements;

	static bool blank(unsigned char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }
	bool word(std::string& w)       // ply_read_word: skip blanks, take the word, consume ONE delimiter
	{
		while (pos < buf.size() && blank(buf[pos])) ++pos;
		if (pos >= buf.size()) return false;
		const size_t b = pos;
		while (pos < buf.size() && !blank(buf[pos])) ++pos;
		w.assign(reinterpret_cast<const char*>(buf.data()) + b, pos - b);
		if (pos < buf.size()) ++pos;
		return true;
	}
	bool skip_line()                // ply_read_line: up to and including the next '\n'
	{
		while (pos < buf.size() && buf[pos] != '\n') ++pos;
		if (pos >= buf.size()) return false;
		++pos; return true;
	}
	bool header()
	{
		std::string w;
		if (!word(w) || w != "format" || !word(w)) return false;
		if (w == "ascii") mode = 0; else if (w == "binary_little_endian") mode = 1; else if (w == "binary_big_endian") mode = 2; else return false;
		if (!word(w) || w != "1.0" || !word(w)) return false;
		while (w != "end_header")
		{
			if (w == "comment" || w == "obj_info") { if (!skip_line() || !word(w)) return false; }
			else if (w == "element")
			{
				PlyElement e;
				if (!word(e.name) || !word(w)) return false;
				int n = 0;
				if (std::sscanf(w.c_str(), "%d", &n) != 1) return false;
				e.count = n;
				if (!word(w)) return false;
				for (;;)
				{
					if (w == "property")
					{
						PlyProperty p;
						if (!word(w)) return false;
						p.type = ply_type(w);
						if (p.type == P_NONE) return false;
						if (p.type == P_LIST)
						{
							if (!word(w)) return false; p.length_type = ply_type(w);
							if (p.length_type == P_NONE) return false;      // (rply accepts "list" here and then indexes past its handler table)
							if (!word(w)) return false; p.value_type = ply_type(w);
							if (p.value_type == P_NONE) return false;
							if (p.length_type == P_LIST || p.value_type == P_LIST) return false;
						}
						if (!word(p.name) || !word(w)) return false;
						e.props.push_back(p);
					}
					else if (w == "comment" || w == "obj_info") { if (!skip_line() || !word(w)) return false; }
					else break;
				}
				elements.push_back(e);
			}
			else return false;          // "Unexpected token"
		}
		return true;
	}
	bool value(PlyType t, double& v)
	{
		if (mode == 0)
		{
			std::string w;
			if (!word(w)) return false;
			char* end = nullptr;
			if (t == P_FLOAT32 || t == P_FLOAT64)
			{
				v = std::strtod(w.c_str(), &end);
				const double lim = t == P_FLOAT32 ? double(FLT_MAX) : DBL_MAX;
				return !*end && !(v < -lim) && !(v > lim);
			}
			v = double(std::strtol(w.c_str(), &end, 10));
			if (*end) return false;
			switch (t)
			{
			case P_INT8:   return v <= CHAR_MAX && v >= CHAR_MIN;
			case P_UINT8:  return v <= UCHAR_MAX && v >= 0;
			case P_INT16:  return v <= SHRT_MAX && v >= SHRT_MIN;
			case P_UINT16: return v <= USHRT_MAX && v >= 0;
			case P_INT32:  return true;
			default:       return v >= 0;
			}
		}
		static const size_t sizes[8] = { 1, 1, 2, 2, 4, 4, 4, 8 };
		const size_t n = sizes[t];
		if (pos + n > buf.size()) return false;
		unsigned char b[8];
		for (size_t i = 0; i < n; ++i) b[i] = buf[pos + (mode == 1 ? i : n - 1 - i)];      // to little endian (the host's order)
		pos += n;
		switch (t)
		{
		case P_INT8:    { signed char x; std::memcpy(&x, b, 1); v = x; break; }
		case P_UINT8:   v = b[0]; break;
		case P_INT16:   { int16_t x; std::memcpy(&x, b, 2); v = x; break; }
		case P_UINT16:  { uint16_t x; std::memcpy(&x, b, 2); v = x; break; }
		case P_INT32:   { int32_t x; std::memcpy(&x, b, 4); v = x; break; }
		case P_UINT32:  { uint32_t x; std::memcpy(&x, b, 4); v = x; break; }
		case P_FLOAT32: { float x; std::memcpy(&x, b, 4); v = x; break; }
		default:        std::memcpy(&v, b, 8); break;
		}
		return true;
	}
	const PlyElement* find(const char* element, const char* property) const      // ply_set_read_cb's lookup: first element / property of that name
	{
		for (const PlyElement& e : elements)
			if (e.name == element)
			{
				for (const PlyProperty& p : e.props) if (p.name == property) return &e;
				return nullptr;
			}
		return nullptr;
	}
};

} // namespace

static void load_ply(const std::string& filename, MeshStorage& mesh)
{
	mesh = MeshStorage();
	PlyReader ply;
	{
		FILE* f = std::fopen(filename.c_str(), "rb");
		if (!f) throw MeshException("Error opening ply file during first pass (" + filename + ")");
		unsigned char tmp[65536]; size_t got;
		while ((got = std::fread(tmp, 1, sizeof(tmp), f)) > 0) ply.buf.insert(ply.buf.end(), tmp, tmp + got);
		std::fclose(f);
	}
	if (ply.buf.size() < 4 || std::memcmp(ply.buf.data(), "ply\n", 4) != 0) throw MeshException("Error opening ply file during first pass (" + filename + ")");
	ply.pos = 4;
	if (!ply.header()) throw MeshException("Error parsing ply header during first pass (" + filename + ")");

	auto count_of = [&](const char* e, const char* p) { const PlyElement* el = ply.find(e, p); return el ? int(el->count) : 0; };
	mesh.num_vertices = count_of("vertex", "x");
	mesh.num_normals = count_of("vertex", "nx");
	mesh.num_texture_coordinates = count_of("vertex", "s");
	if (mesh.num_texture_coordinates == 0) mesh.num_texture_coordinates = count_of("vertex", "u");
	mesh.num_triangles = count_of("face", "vertex_indices");
	if (mesh.num_vertices < 0 || mesh.num_triangles < 0) throw MeshException("Error parsing ply header during first pass (" + filename + ")");

	mesh.vertex_data.assign(size_t(mesh.num_vertices) * 4, 0.0f);
	mesh.normal_data.assign(size_t(mesh.num_normals) * 3, 0.0f);
	mesh.texture_data.assign(size_t(mesh.num_texture_coordinates) * 2, 0.0f);
	mesh.vertex_indices.assign(size_t(mesh.num_triangles) * 4, 0);
	if (mesh.num_normals) mesh.normal_indices.assign(size_t(mesh.num_triangles) * 4, 0);
	if (mesh.num_texture_coordinates) mesh.texture_indices.assign(size_t(mesh.num_triangles) * 4, 0);

	// the callbacks (MeshBase.cpp:254-340) are attached to the FIRST element called "vertex" / "face" and keep running cursors
	const PlyElement* vertex_el = nullptr; const PlyElement* face_el = nullptr;
	for (const PlyElement& e : ply.elements) { if (!vertex_el && e.name == "vertex") vertex_el = &e; if (!face_el && e.name == "face") face_el = &e; }
	int cur_v = 0, cur_n = 0, cur_t = 0, cur_tri = 0;
	auto overflow = [&]() { throw MeshException("Error parsing ply file (" + filename + ")"); };
	auto vertex_cb = [&](int coord, double dv)
	{
		const float value = float(dv);
		switch (coord)
		{
		case 0: case 1: case 2:
			if (cur_v >= mesh.num_vertices) overflow();
			mesh.vertex_data[size_t(cur_v) * 4 + coord] = value; if (coord == 2) ++cur_v; break;
		case 3: case 4: case 5:
			if (cur_n >= mesh.num_normals) overflow();
			mesh.normal_data[size_t(cur_n) * 3 + (coord - 3)] = value; if (coord == 5) ++cur_n; break;
		default:
			if (cur_t >= mesh.num_texture_coordinates) overflow();
			mesh.texture_data[size_t(cur_t) * 2 + (coord - 6)] = value; if (coord == 7) ++cur_t; break;
		}
	};
	for (const PlyElement& e : ply.elements)
	{
		for (long j = 0; j < e.count; ++j)
			for (const PlyProperty& p : e.props)
			{
				if (p.type != P_LIST)
				{
					double v;
					if (!ply.value(p.type, v)) throw MeshException("Error parsing ply file (" + filename + ")");
					if (&e != vertex_el) continue;
					int coord = -1;
					if (p.name == "x") coord = 0; else if (p.name == "y") coord = 1; else if (p.name == "z") coord = 2;
					else if (mesh.num_normals && p.name == "nx") coord = 3; else if (mesh.num_normals && p.name == "ny") coord = 4;
					else if (mesh.num_normals && p.name == "nz") coord = 5;
					else if (mesh.num_texture_coordinates && (p.name == "s" || p.name == "u")) coord = 6;
					else if (mesh.num_texture_coordinates && (p.name == "t" || p.name == "v")) coord = 7;
					if (coord >= 0) vertex_cb(coord, v);
				}
				else
				{
					double len;
					if (!ply.value(p.length_type, len)) throw MeshException("Error parsing ply file (" + filename + ")");
					const bool is_face = &e == face_el && p.name == "vertex_indices";
					for (int l = 0; l < int(len); ++l)
					{
						double v;
						if (!ply.value(p.value_type, v)) throw MeshException("Error parsing ply file (" + filename + ")");
						if (!is_face || l > 2) continue;
						if (cur_tri >= mesh.num_triangles) overflow();
						const int value = int(v);
						mesh.vertex_indices[size_t(cur_tri) * 4 + l] = value;
						if (mesh.num_normals) mesh.normal_indices[size_t(cur_tri) * 4 + l] = value;
						if (mesh.num_texture_coordinates) mesh.texture_indices[size_t(cur_tri) * 4 + l] = value;
						if (l == 2) ++cur_tri;
					}
				}
			}
	}
	mesh.material_indices.assign(size_t(mesh.num_triangles), 0);
	mesh.group_names.push_back("null-group");
	mesh.group_offsets.clear(); mesh.group_offsets.push_back(0); mesh.group_offsets.push_back(mesh.num_triangles);
	const MeshMaterialParams def;                                   // insertDefaultMaterial = true (MeshBase.h:201)
	mesh.materials.push_back(make_material(mesh, def, true));
	mesh.material_names.push_back(def.name);
}

// MeshBase::loadModel (src/mesh/MeshBase.cpp:446-460): dispatch on the (case-sensitive) extension
void loadModel(const std::string& filename, MeshStorage& mesh)
{
	const std::string::size_type dot = filename.find_last_of('.');
	const std::string ext = dot != std::string::npos ? filename.substr(dot + 1) : std::string();
	if (ext == "obj") load_obj(filename, mesh);
	else if (ext == "ply") load_ply(filename, mesh);
	else throw MeshException("Unrecognized model file extension (" + filename + ")");
}

void loadMaterials(const std::string& filename, MeshStorage& mesh)
{
	std::vector<MeshMaterialParams> staging;
	if (!read_mtl(filename, staging)) return;
	for (const MeshMaterialParams& p : staging)
	{
		mesh.materials.push_back(make_material(mesh, p, false));    // bump maps dropped here, src/mesh/MeshStorage.cpp:229
		mesh.material_names.push_back(p.name);
	}
}

// ---- mesh operators (src/mesh/MeshStorage.cpp:449-648) --------------------------------------------------------------------------
void add_per_triangle_normals(MeshStorage& mesh)
{
	const int nt = mesh.num_triangles;
	mesh.num_normals = nt;
	mesh.normal_indices.assign(size_t(nt) * 4, 0);
	mesh.normal_data.assign(size_t(nt) * 3, 0.0f);
	for (int t = 0; t < nt; ++t)
	{
		mesh.normal_indices[size_t(t) * 4 + 0] = mesh.normal_indices[size_t(t) * 4 + 1] = mesh.normal_indices[size_t(t) * 4 + 2] = t;
		const int* tri = &mesh.vertex_indices[size_t(t) * 4];
		auto vtx = [&](int i) { const float* p = &mesh.vertex_data[size_t(i) * 4]; return fpt::mk3(p[0], p[1], p[2]); };
		const fpt::f3 ng = fpt::normalize(fpt::cross(vtx(tri[0]) - vtx(tri[2]), vtx(tri[1]) - vtx(tri[2])));
		mesh.normal_data[size_t(t) * 3 + 0] = ng.x; mesh.normal_data[size_t(t) * 3 + 1] = ng.y; mesh.normal_data[size_t(t) * 3 + 2] = ng.z;
	}
}

void add_per_triangle_texture_coordinates(MeshStorage& mesh)
{
	const int nt = mesh.num_triangles;
	mesh.num_texture_coordinates = 3;
	mesh.texture_indices.assign(size_t(nt) * 4, 0);
	for (int t = 0; t < nt; ++t) { mesh.texture_indices[size_t(t) * 4 + 1] = 1; mesh.texture_indices[size_t(t) * 4 + 2] = 2; }
	const float tri[6] = { 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 1.0f };
	mesh.texture_data.assign(tri, tri + 6);
}

void merge(MeshStorage& mesh, const MeshStorage& other_in)
{
	const int num_materials = int(mesh.materials.size());
	MeshStorage copy;
	const MeshStorage* other = &other_in;
	if ((mesh.num_normals > 0) != (other->num_normals > 0))
	{
		if (mesh.num_normals == 0) { if (mesh.num_triangles) add_per_triangle_normals(mesh); }
		else { copy = *other; other = &copy; add_per_triangle_normals(copy); }
	}
	if ((mesh.num_texture_coordinates > 0) != (other->num_texture_coordinates > 0))
	{
		if (mesh.num_texture_coordinates == 0) { if (mesh.num_triangles) add_per_triangle_texture_coordinates(mesh); }
		else { if (other == &other_in) { copy = *other; other = &copy; } add_per_triangle_texture_coordinates(copy); }
	}
	auto append = [](std::vector<int>& dst, const std::vector<int>& src, int offset)
	{
		const size_t base = dst.size();
		dst.resize(base + src.size());
		for (size_t i = 0; i < src.size(); ++i) dst[base + i] = src[i] + ((i % 4) < 3 ? offset : 0);
	};
	append(mesh.vertex_indices, other->vertex_indices, mesh.num_vertices);
	append(mesh.normal_indices, other->normal_indices, mesh.num_normals);
	append(mesh.texture_indices, other->texture_indices, mesh.num_texture_coordinates);
	for (int m : other->material_indices) mesh.material_indices.push_back(m + num_materials);
	mesh.vertex_data.insert(mesh.vertex_data.end(), other->vertex_data.begin(), other->vertex_data.end());
	mesh.normal_data.insert(mesh.normal_data.end(), other->normal_data.begin(), other->normal_data.end());
	mesh.texture_data.insert(mesh.texture_data.end(), other->texture_data.begin(), other->texture_data.end());
	if (mesh.group_offsets.empty()) mesh.group_offsets.push_back(0);
	mesh.group_offsets.pop_back();
	for (size_t i = 0; i < other->group_offsets.size(); ++i) mesh.group_offsets.push_back(mesh.num_triangles + other->group_offsets[i]);
	if (other->group_offsets.empty()) mesh.group_offsets.push_back(mesh.num_triangles + other->num_triangles);
	mesh.group_names.insert(mesh.group_names.end(), other->group_names.begin(), other->group_names.end());
	mesh.num_vertices += other->num_vertices;
	mesh.num_normals += other->num_normals;
	mesh.num_texture_coordinates += other->num_texture_coordinates;
	mesh.num_triangles += other->num_triangles;
	for (size_t i = 0; i < other->materials.size(); ++i)
	{
		fpt_material m = other->materials[i];
		auto relink = [&](fpt_texture_ref& r)
		{
			if (r.texture == 0xFFFFFFFFu) return;
			MeshTextureMap t; t.name = other->textures[r.texture]; t.scaling[0] = r.scaling[0]; t.scaling[1] = r.scaling[1];
			r.texture = insert_texture(mesh, t).texture;
		};
		relink(m.ambient_map); relink(m.diffuse_map); relink(m.diffuse_trans_map); relink(m.specular_map); relink(m.emissive_map); relink(m.bump_map);
		mesh.materials.push_back(m);
		mesh.material_names.push_back(other->material_names[i]);
	}
}

namespace {
struct Mat4 { float m[4][4]; };
Mat4 identity() { Mat4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.0f : 0.0f; return r; }
Mat4 mul(const Mat4& a, const Mat4& b)
{
	Mat4 r;
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
	{
		float s = 0.0f;
		for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
		r.m[i][j] = s;
	}
	return r;
}
// cofactor inverse (contrib/cugar/linalg/matrix_inline.h:292-360)
bool invert(const Mat4& a, Mat4& r)
{
	float c[4][4];
	auto det3 = [&](int r0, int r1, int r2, int c0, int c1, int c2)
	{
		return a.m[r0][c0] * (a.m[r1][c1] * a.m[r2][c2] - a.m[r1][c2] * a.m[r2][c1])
		     - a.m[r0][c1] * (a.m[r1][c0] * a.m[r2][c2] - a.m[r1][c2] * a.m[r2][c0])
		     + a.m[r0][c2] * (a.m[r1][c0] * a.m[r2][c1] - a.m[r1][c1] * a.m[r2][c0]);
	};
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
	{
		int rr[3], cc[3], n = 0, k = 0;
		for (int x = 0; x < 4; ++x) if (x != i) rr[n++] = x;
		for (int x = 0; x < 4; ++x) if (x != j) cc[k++] = x;
		c[i][j] = (((i + j) & 1) ? -1.0f : 1.0f) * det3(rr[0], rr[1], rr[2], cc[0], cc[1], cc[2]);
	}
	const float det = a.m[0][0] * c[0][0] + a.m[0][1] * c[0][1] + a.m[0][2] * c[0][2] + a.m[0][3] * c[0][3];
	if (det == 0.0f) return false;
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = c[j][i] / det;
	return true;
}
} // namespace

// points by M, normals by the inverse transpose, NOT re-normalised (src/mesh/MeshStorage.cpp:623-639)
void transform(MeshStorage& mesh, const float mat[16])
{
	Mat4 M, Ni, N;
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) M.m[i][j] = mat[i * 4 + j];
	if (!invert(M, Ni)) Ni = identity();
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) N.m[i][j] = Ni.m[j][i];
	for (int i = 0; i < mesh.num_vertices; ++i)
	{
		float* p = &mesh.vertex_data[size_t(i) * 4];
		const float v[4] = { p[0], p[1], p[2], 1.0f };
		float r[3];
		for (int k = 0; k < 3; ++k) r[k] = ((M.m[k][0] * v[0] + M.m[k][1] * v[1]) + M.m[k][2] * v[2]) + M.m[k][3] * v[3];
		p[0] = r[0]; p[1] = r[1]; p[2] = r[2];
	}
	for (int i = 0; i < mesh.num_normals; ++i)
	{
		float* p = &mesh.normal_data[size_t(i) * 3];
		const float v[3] = { p[0], p[1], p[2] };
		for (int k = 0; k < 3; ++k) p[k] = (N.m[k][0] * v[0] + N.m[k][1] * v[1]) + N.m[k][2] * v[2];
	}
}

// compress_tex (src/mesh/MeshStorage.cpp:270-299) + compress_tex_coord (src/mesh/MeshCompression.h:36-48)
void MeshStorage::compress_tex()
{
	if (!num_texture_coordinates) return;
	float lo[2] = { 1.0e16f, 1.0e16f }, hi[2] = { -1.0e16f, -1.0e16f };
	for (int i = 0; i < num_texture_coordinates; ++i) for (int c = 0; c < 2; ++c)
	{
		lo[c] = std::min(lo[c], texture_data[size_t(i) * 2 + c]);
		hi[c] = std::max(hi[c], texture_data[size_t(i) * 2 + c]);
	}
	tex_bias[0] = lo[0]; tex_bias[1] = lo[1]; tex_scale[0] = hi[0] - lo[0]; tex_scale[1] = hi[1] - lo[1];
	texture_indices_comp.assign(size_t(num_triangles) * 4, -1);
	for (int t = 0; t < num_triangles; ++t) for (int c = 0; c < 3; ++c)
	{
		const int idx = texture_indices[size_t(t) * 4 + c];
		if (idx < 0) continue;
		// a degenerate axis (every u, or every v, equal: scale 0) would give 0/0 = NaN coordinates; the offset from the bias is 0 there
		const float u = tex_scale[0] != 0.0f ? (texture_data[size_t(idx) * 2 + 0] - tex_bias[0]) / tex_scale[0] : 0.0f;
		const float v = tex_scale[1] != 0.0f ? (texture_data[size_t(idx) * 2 + 1] - tex_bias[1]) / tex_scale[1] : 0.0f;
		texture_indices_comp[size_t(t) * 4 + c] = int(fpt::float_to_half_bits(u) | (fpt::float_to_half_bits(v) << 16));
	}
}

// unify_vertex_attributes (src/mesh/MeshStorage.cpp:651-840), with compress_normals (:246-266) folded in: one vertex per distinct
// (position, normal, texcoord) index triple in first-seen order; triangles without normals use their geometric normal
void unify_vertex_attributes(MeshStorage& mesh)
{
	typedef std::tuple<int, int, int> Key;
	// first-seen numbering of the distinct triples.  The reference keeps them in an ordered map; the numbering only depends on the order of
	// insertion (kept in `vertices`), so an open-addressing table gives the same result at a fraction of the time on multi-million-triangle scenes.
	const int nt = mesh.num_triangles;
	const bool has_n = !mesh.normal_indices.empty(), has_t = !mesh.texture_indices.empty();
	auto key = [&](int t, int c)
	{
		const int n = has_n ? mesh.normal_indices[size_t(t) * 4 + c] : -1;
		return Key(mesh.vertex_indices[size_t(t) * 4 + c], n >= 0 ? n : -t - 1, has_t ? mesh.texture_indices[size_t(t) * 4 + c] : -1);
	};
	size_t cap = 16; while (cap < size_t(nt) * 6 + 16) cap <<= 1;
	std::vector<uint32> table(cap, 0xFFFFFFFFu);            // slot -> index into `vertices`
	std::vector<Key> vertices;
	std::vector<int> nv(size_t(nt) * 4);                    // the new index stream
	auto hash = [](const Key& k) {
		uint64_t h = uint64_t(uint32_t(std::get<0>(k))) * 0x9E3779B97F4A7C15ull;
		h ^= (uint64_t(uint32_t(std::get<1>(k))) + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full; h ^= h >> 29;
		h += uint64_t(uint32_t(std::get<2>(k))) * 0x165667B19E3779F9ull; h ^= h >> 32;
		return size_t(h); };
	for (int t = 0; t < nt; ++t)
	{
		for (int c = 0; c < 3; ++c)
		{
			const Key k = key(t, c);
			size_t slot = hash(k) & (cap - 1);
			while (table[slot] != 0xFFFFFFFFu && vertices[table[slot]] != k) slot = (slot + 1) & (cap - 1);
			if (table[slot] == 0xFFFFFFFFu) { table[slot] = uint32(vertices.size()); vertices.push_back(k); }
			nv[size_t(t) * 4 + c] = int(table[slot]);
		}
		nv[size_t(t) * 4 + 3] = mesh.vertex_indices[size_t(t) * 4 + 3];
	}
	std::vector<uint32>().swap(table);
	std::vector<float> vdata(vertices.size() * 4), ndata(vertices.size() * 3), tdata(vertices.size() * 2, 0.0f);
	for (size_t i = 0; i < vertices.size(); ++i)
	{
		const int v = std::get<0>(vertices[i]), n = std::get<1>(vertices[i]), tx = std::get<2>(vertices[i]);
		for (int c = 0; c < 3; ++c) vdata[i * 4 + c] = mesh.vertex_data[size_t(v) * 4 + c];
		if (tx >= 0) { tdata[i * 2] = mesh.texture_data[size_t(tx) * 2]; tdata[i * 2 + 1] = mesh.texture_data[size_t(tx) * 2 + 1]; }
		fpt::f3 nn;
		if (n >= 0) nn = fpt::mk3(mesh.normal_data[size_t(n) * 3], mesh.normal_data[size_t(n) * 3 + 1], mesh.normal_data[size_t(n) * 3 + 2]);
		else
		{
			const int* tri = &mesh.vertex_indices[size_t(-n - 1) * 4];
			auto vtx = [&](int k) { const float* p = &mesh.vertex_data[size_t(k) * 4]; return fpt::mk3(p[0], p[1], p[2]); };
			nn = fpt::normalize(fpt::cross(vtx(tri[0]) - vtx(tri[2]), vtx(tri[1]) - vtx(tri[2])));
		}
		ndata[i * 3] = nn.x; ndata[i * 3 + 1] = nn.y; ndata[i * 3 + 2] = nn.z;
		vdata[i * 4 + 3] = fpt::as_f32(fpt::pack_normal(nn));
	}
	// the new index stream replaces the old ones, the unified attribute arrays are swapped in
	mesh.vertex_indices = nv;
	if (has_n) { mesh.normal_indices = nv; }
	if (has_t) { mesh.texture_indices = nv; }
	mesh.vertex_data.swap(vdata); mesh.normal_data.swap(ndata); mesh.texture_data.swap(tdata);
	mesh.num_vertices = mesh.num_normals = mesh.num_texture_coordinates = int(vertices.size());
}

void apply_material_flags(MeshStorage& mesh)
{
	for (int t = 0; t < mesh.num_triangles; ++t)
	{
		const int m = mesh.material_indices[size_t(t)];
		if (m > -1) mesh.vertex_indices[size_t(t) * 4 + 3] = mesh.materials[size_t(m)].flags;
	}
}

// ---- files ----------------------------------------------------------------------------------------------------------------------
namespace {
std::string extract_path(const std::string& filename)       // src/files.cpp:34-50
{
	const size_t k = filename.find_last_of("/\\");
	return k == std::string::npos ? std::string() : filename.substr(0, k);
}
bool find_file(std::string& name, const std::vector<std::string>& dirs)      // src/files.cpp:70-86
{
	for (const std::string& d : dirs)
	{
		const std::string full = d + "/" + name;
		FILE* f = std::fopen((d.empty() ? name : full).c_str(), "r");
		if (f) { std::fclose(f); if (!d.empty()) name = full; return true; }
	}
	return false;
}
bool ends_with(const std::string& s, const char* suffix)
{
	const size_t n = std::strlen(suffix);
	return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}
void finish_camera(fpt_camera& c)
{
	const fpt::f3 d = fpt::normalize(fpt::cross(fpt::mk3(c.aim[0] - c.eye[0], c.aim[1] - c.eye[1], c.aim[2] - c.eye[2]), fpt::mk3(c.up[0], c.up[1], c.up[2])));
	c.dx[0] = d.x; c.dx[1] = d.y; c.dx[2] = d.z;
}
fpt_camera default_camera()                                  // Camera::Camera (src/camera.h:53-59)
{
	fpt_camera c; std::memset(&c, 0, sizeof(c));
	c.eye[2] = -1.0f; c.up[1] = 1.0f; c.dx[0] = 1.0f; c.fov = float(M_PI) / 4.0f;
	return c;
}
} // namespace

bool load_camera_file(const char* filename, fpt_camera& c)
{
	Scanner sc;
	if (!sc.open(filename)) return false;
	for (int k = 0; k < 3; ++k) sc.read_float(c.eye[k]);
	for (int k = 0; k < 3; ++k) sc.read_float(c.aim[k]);
	for (int k = 0; k < 3; ++k) sc.read_float(c.up[k]);
	sc.read_float(c.fov);
	finish_camera(c);
	return true;
}

// ---- .fa scene scripts (src/mesh/fermat_loader.cpp:46-360) ------------------------------------------------------------------------
void load_scene(const char* filename, MeshStorage& mesh, std::vector<fpt_camera>& cameras, std::vector<fpt_dir_light>& dir_lights,
                std::vector<std::string>& dirs, std::vector<std::string>& scene_dirs)
{
	const std::string fname(filename);
	if (!ends_with(fname, ".fa"))
	{
		if (ends_with(fname, ".obj") || ends_with(fname, ".ply")) { loadModel(fname, mesh); return; }
		throw MeshException("unsupported scene format (this build reads .fa, .obj and .ply): " + fname);
	}
	Scanner sc;
	if (!sc.open(fname)) throw MeshException("unable to open file: " + fname);
	int default_material = -1;
	std::vector<Mat4> stack(1, identity());
	std::string cmd;
	auto need = [&](float& v, const char* what) { if (!sc.read_float(v)) throw MeshException(std::string(what) + ": insufficient number of arguments"); };
	while (sc.token(cmd))
	{
		if (cmd[0] == '#') sc.line();
		else if (cmd == "Begin") stack.push_back(stack.back());
		else if (cmd == "End") { if (stack.size() > 1) stack.pop_back(); }
		else if (cmd == "Transform")
		{
			Mat4 m;
			for (int i = 0; i < 16; ++i) need(m.m[i / 4][i % 4], "Transform");
			stack.back() = mul(m, stack.back());
		}
		else if (cmd == "Translate")
		{
			Mat4 m = identity();
			need(m.m[0][3], "Translate"); need(m.m[1][3], "Translate"); need(m.m[2][3], "Translate");
			stack.back() = mul(m, stack.back());
		}
		else if (cmd == "Scale")
		{
			Mat4 m = identity();
			need(m.m[0][0], "Scale"); need(m.m[1][1], "Scale"); need(m.m[2][2], "Scale");
			stack.back() = mul(m, stack.back());
		}
		else if (cmd == "RotateX" || cmd == "RotateY" || cmd == "RotateZ")
		{
			float angle = 0.0f; sc.read_float(angle);
			const float q = angle * float(M_PI) / 180.0f, s = std::sin(q), c = std::cos(q);
			Mat4 m = identity();                                   // contrib/cugar/linalg/matrix_inline.h:615-681
			if (cmd[6] == 'X') { m.m[1][1] = m.m[2][2] = c; m.m[1][2] = -s; m.m[2][1] = s; }
			else if (cmd[6] == 'Y') { m.m[0][0] = m.m[2][2] = c; m.m[2][0] = -s; m.m[0][2] = s; }
			else { m.m[0][0] = m.m[1][1] = c; m.m[1][0] = s; m.m[0][1] = -s; }
			stack.back() = mul(m, stack.back());
		}
		else if (cmd == "LoadScene" || cmd == "LoadMesh")
		{
			std::string name; sc.token(name);
			if (!find_file(name, dirs)) throw MeshException("unable to find file \"" + name + "\"");
			scene_dirs.push_back(extract_path(name));
			MeshStorage other;
			load_scene(name.c_str(), other, cameras, dir_lights, dirs, scene_dirs);
			transform(other, &stack.back().m[0][0]);
			const int triangle_offset = mesh.num_triangles;
			const int num_materials = int(mesh.materials.size());
			merge(mesh, other);
			if (default_material != -1)                             // the merged mesh's default material (its index 0) is replaced
				for (int i = 0; i < other.num_triangles; ++i)
					if (mesh.material_indices[size_t(triangle_offset + i)] == num_materials) mesh.material_indices[size_t(triangle_offset + i)] = default_material;
		}
		else if (cmd == "LoadMaterials")
		{
			std::string name; sc.token(name);
			if (!find_file(name, dirs)) throw MeshException("unable to find file \"" + name + "\"");
			loadMaterials(name, mesh);
		}
		else if (cmd == "SetMaterial")
		{
			std::string name; sc.token(name);
			for (int i = int(mesh.materials.size()) - 1; i >= 0; --i) if (mesh.material_names[size_t(i)] == name) { default_material = i; break; }
		}
		else if (cmd == "Camera")
		{
			std::istringstream ss(sc.line());
			std::string type; ss >> type;
			if (type != "persp") { std::fprintf(stderr, "warning: unsupported camera type \"%s\", in file %s\n", type.c_str(), filename); continue; }
			fpt_camera cam = default_camera();
			std::string p;
			while (ss >> p)
			{
				bool ok = true;
				if (p == "eye") ok = bool(ss >> cam.eye[0] >> cam.eye[1] >> cam.eye[2]);
				else if (p == "aim") ok = bool(ss >> cam.aim[0] >> cam.aim[1] >> cam.aim[2]);
				else if (p == "up") ok = bool(ss >> cam.up[0] >> cam.up[1] >> cam.up[2]);
				else if (p == "fov") ok = bool(ss >> cam.fov);
				else { std::fprintf(stderr, "warning: unsupported Camera parameter \"%s\", in file %s\n", p.c_str(), filename); break; }
				if (!ok) { std::fprintf(stderr, "warning: badly formatted value for Camera parameter \"%s\", in file %s\n", p.c_str(), filename); break; }
			}
			finish_camera(cam);
			cameras.push_back(cam);
		}
		else if (cmd == "DirectionalLight")
		{
			std::istringstream ss(sc.line());
			fpt_dir_light l; std::memset(&l, 0, sizeof(l));
			std::string p;
			while (ss >> p)
			{
				bool ok = true;
				if (p == "dir" || p == "direction")
				{
					ok = bool(ss >> l.dir[0] >> l.dir[1] >> l.dir[2]);
					const fpt::f3 d = fpt::normalize(fpt::mk3(l.dir[0], l.dir[1], l.dir[2]));
					l.dir[0] = d.x; l.dir[1] = d.y; l.dir[2] = d.z;
				}
				else if (p == "color") ok = bool(ss >> l.color[0] >> l.color[1] >> l.color[2]);
				else { std::fprintf(stderr, "warning: unsupported DirectionalLight parameter \"%s\", in file %s\n", p.c_str(), filename); break; }
				if (!ok) { std::fprintf(stderr, "warning: badly formatted value for DirectionalLight parameter \"%s\", in file %s\n", p.c_str(), filename); break; }
			}
			dir_lights.push_back(l);
		}
	}
}

// ---- images (contrib/cugar/image/tga.cpp, pfm.cpp) ------------------------------------------------------------------------------
unsigned char* load_tga(const char* filename, int* width, int* height, int* bits)
{
	FILE* fp = std::fopen(filename, "rb");
	if (!fp) return nullptr;
	unsigned char h[18];
	if (std::fread(h, 1, 18, fp) != 18) { std::fclose(fp); return nullptr; }
	const int identsize = h[0], cmaptype = h[1], imagetype = h[2], cmapstart = h[3] | (h[4] << 8), cmaplen = h[5] | (h[6] << 8), cmapbits = h[7];
	const int w = h[12] | (h[13] << 8), ht = h[14] | (h[15] << 8), bpp = h[16];
	for (int i = 0; i < identsize; ++i) std::fgetc(fp);
	const size_t n = size_t(w) * size_t(ht);
	unsigned char* pix = nullptr;
	if (imagetype == 1)                                        // colour-mapped, 24-bit palette, 8-bit indices
	{
		if (cmaptype != 1 || cmapbits != 24 || bpp != 8) { std::fclose(fp); return nullptr; }
		std::vector<unsigned char> map(size_t(3) * cmaplen), idx(n);
		if (std::fread(map.data(), 1, map.size(), fp) != map.size() || std::fread(idx.data(), 1, n, fp) != n) { std::fclose(fp); return nullptr; }
		pix = new unsigned char[n * 3];
		for (size_t i = 0; i < n; ++i)
		{
			// palette entries are numbered from the header's first-entry index; an index outside the stored palette is a corrupt file
			const int ci = int(idx[i]) - cmapstart;
			if (ci < 0 || ci >= cmaplen) { delete[] pix; std::fclose(fp); return nullptr; }
			pix[i * 3 + 0] = map[ci * 3 + 2]; pix[i * 3 + 1] = map[ci * 3 + 1]; pix[i * 3 + 2] = map[ci * 3 + 0];
		}
		*bits = 24;
	}
	else if ((imagetype == 2 || imagetype == 10) && (bpp == 24 || bpp == 32))
	{
		const size_t nb = size_t(bpp) >> 3;
		pix = new unsigned char[n * nb]();                     // zero-initialised: a truncated RLE stream leaves the remainder black, never garbage
		if (imagetype == 2) { if (std::fread(pix, 1, n * nb, fp) != n * nb) { delete[] pix; std::fclose(fp); return nullptr; } }
		else                                                   // run-length packets (not read by the reference; accepted here)
		{
			size_t i = 0;
			bool truncated = false;
			while (i < n)
			{
				const int c = std::fgetc(fp);
				if (c == EOF) { truncated = true; break; }
				const size_t run = size_t(c & 0x7f) + 1;
				if (c & 0x80)
				{
					unsigned char px[4];
					if (std::fread(px, 1, nb, fp) != nb) { truncated = true; break; }
					for (size_t k = 0; k < run && i < n; ++k, ++i) std::memcpy(pix + i * nb, px, nb);
				}
				else { const size_t m = std::min(run, n - i); if (std::fread(pix + i * nb, 1, m * nb, fp) != m * nb) { truncated = true; break; } i += m; }
			}
			if (truncated) { delete[] pix; std::fclose(fp); return nullptr; }     // as for a short uncompressed file
		}
		for (size_t i = 0; i < n; ++i) std::swap(pix[i * nb + 0], pix[i * nb + 2]);      // BGR -> RGB
		// rows stay in stored order: the reference's reader is a raw read that ignores the descriptor's origin bit (contrib/cugar/image/tga.cpp)
		*bits = bpp;
	}
	std::fclose(fp);
	if (pix) { *width = w; *height = ht; }
	return pix;
}

bool write_tga(const char* filename, int width, int height, const unsigned char* pixdata, int channels)
{
	FILE* fp = std::fopen(filename, "wb");
	if (!fp) return false;
	unsigned char h[18]; std::memset(h, 0, 18);
	h[2] = 2; h[12] = (unsigned char)(width & 0xff); h[13] = (unsigned char)(width >> 8); h[14] = (unsigned char)(height & 0xff); h[15] = (unsigned char)(height >> 8); h[16] = 24;
	std::fwrite(h, 1, 18, fp);
	std::vector<unsigned char> row(size_t(width) * 3);
	for (int y = 0; y < height; ++y)
	{
		for (int x = 0; x < width; ++x)
		{
			const unsigned char* p = pixdata + (size_t(y) * width + x) * channels;
			row[size_t(x) * 3 + 0] = p[2]; row[size_t(x) * 3 + 1] = p[1]; row[size_t(x) * 3 + 2] = p[0];
		}
		std::fwrite(row.data(), 1, row.size(), fp);
	}
	std::fclose(fp);
	return true;
}

float* load_pfm(const char* filename, uint32* xres, uint32* yres)
{
	FILE* f = std::fopen(filename, "rb");
	if (!f) return nullptr;
	char magic[8] = { 0 }; int w = 0, h = 0; float scale = 1.0f;
	if (std::fscanf(f, "%7s %d %d %f", magic, &w, &h, &scale) != 4 || w <= 0 || h <= 0) { std::fclose(f); return nullptr; }
	std::fgetc(f);                                             // the single whitespace after the scale
	const int nc = std::strcmp(magic, "PF") == 0 ? 3 : (std::strcmp(magic, "Pf") == 0 ? 1 : 0);
	if (!nc) { std::fclose(f); return nullptr; }
	std::vector<float> raw(size_t(w) * h * nc);
	if (std::fread(raw.data(), 4, raw.size(), f) != raw.size()) { std::fclose(f); return nullptr; }
	std::fclose(f);
	if (scale > 0.0f)                                          // big-endian payload
		for (float& v : raw) { uint32_t u = fpt::as_u32(v); u = (u >> 24) | ((u >> 8) & 0xff00u) | ((u << 8) & 0xff0000u) | (u << 24); v = fpt::as_f32(u); }
	float* rgb = new float[size_t(w) * h * 3];
	for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) for (int c = 0; c < 3; ++c)       // stored bottom-up
		rgb[(size_t(y) * w + x) * 3 + c] = raw[(size_t(h - 1 - y) * w + x) * nc + (nc == 3 ? c : 0)];
	*xres = uint32(w); *yres = uint32(h);
	return rgb;
}

// ---- HostScene : the scene half of RenderingContextImpl::init (src/renderer.cu:646-870) -------------------------------------------
void HostScene::load(const char* filename, const char* data_directory)
{
	data_dir = data_directory ? data_directory : ".";
	{
		glossy_reflectance.resize(size_t(32) * 32 * 32 * 32);
		const std::string path = data_dir + "/glossy_reflectance.dat";
		FILE* f = std::fopen(path.c_str(), "rb");
		if (!f) throw MeshException("error opening " + path);
		const size_t got = std::fread(glossy_reflectance.data(), sizeof(float), glossy_reflectance.size(), f);
		std::fclose(f);
		if (got != glossy_reflectance.size()) throw MeshException("error loading " + path);
	}
	std::vector<std::string> scene_dirs;
	scene_dirs.push_back("");
	scene_dirs.push_back(extract_path(filename));
	std::vector<std::string> dirs = scene_dirs;
	load_scene(filename, mesh, cameras, dir_lights, dirs, scene_dirs);
	mesh.compress_tex();                   // compress_normals is folded into unify_vertex_attributes
	unify_vertex_attributes(mesh);
	apply_material_flags(mesh);
	for (int c = 0; c < 3; ++c) { bbox[c] = 1.0e16f; bbox[3 + c] = -1.0e16f; }
	for (int i = 0; i < mesh.num_vertices; ++i) for (int c = 0; c < 3; ++c)
	{
		bbox[c] = std::min(bbox[c], mesh.vertex_data[size_t(i) * 4 + c]);
		bbox[3 + c] = std::max(bbox[3 + c], mesh.vertex_data[size_t(i) * 4 + c]);
	}
	for (int i = 0; i < mesh.num_triangles; ++i)
	{
		const int m = mesh.material_indices[size_t(i)];
		if (m < 0 || m >= int(mesh.materials.size())) throw MeshException("material index out of range");
	}
	// textures: float4 texels = bytes/255 (TGA) or raw floats (PFM), alpha 0 (src/renderer.cu:805-850)
	texels.assign(mesh.textures.size(), std::vector<float>());
	texture_views.assign(mesh.textures.size(), fpt_texture{ nullptr, 0, 0 });
	for (size_t i = 0; i < mesh.textures.size(); ++i)
	{
		std::string name = mesh.textures[i];
		std::replace(name.begin(), name.end(), '\\', '/');
		if (!find_file(name, scene_dirs)) { std::fprintf(stderr, "warning: unable to find texture %s\n", name.c_str()); continue; }
		if (ends_with(name, ".tga"))
		{
			int w = 0, h = 0, bits = 0;
			unsigned char* rgb = load_tga(name.c_str(), &w, &h, &bits);
			if (!rgb) { std::fprintf(stderr, "warning: unable to load texture %s\n", name.c_str()); continue; }
			const size_t nb = size_t(bits) >> 3;
			texels[i].resize(size_t(w) * h * 4);
			for (size_t p = 0; p < size_t(w) * h; ++p)
			{
				for (int c = 0; c < 3; ++c) texels[i][p * 4 + c] = float(rgb[p * nb + c]) / 255.0f;
				texels[i][p * 4 + 3] = 0.0f;
			}
			delete[] rgb;
			texture_views[i].res_x = uint32(w); texture_views[i].res_y = uint32(h);
		}
		else if (ends_with(name, ".pfm"))
		{
			uint32 w = 0, h = 0;
			float* rgb = load_pfm(name.c_str(), &w, &h);
			if (!rgb) { std::fprintf(stderr, "warning: unable to load texture %s\n", name.c_str()); continue; }
			texels[i].resize(size_t(w) * h * 4);
			for (size_t p = 0; p < size_t(w) * h; ++p) { for (int c = 0; c < 3; ++c) texels[i][p * 4 + c] = rgb[p * 3 + c]; texels[i][p * 4 + 3] = 0.0f; }
			delete[] rgb;
			texture_views[i].res_x = w; texture_views[i].res_y = h;
		}
		else std::fprintf(stderr, "warning: unsupported texture format %s\n", name.c_str());
	}
	for (size_t i = 0; i < texels.size(); ++i) texture_views[i].texels = texels[i].empty() ? nullptr : texels[i].data();
}

SceneArrays HostScene::arrays(const fpt_camera* override_camera) const
{
	SceneArrays a; std::memset(&a, 0, sizeof(a));
	a.mesh.num_triangles = mesh.num_triangles; a.mesh.num_vertices = mesh.num_vertices; a.mesh.num_materials = int(mesh.materials.size());
	a.mesh.vertex_indices = mesh.vertex_indices.data();
	a.mesh.vertex_data = mesh.vertex_data.data();
	a.mesh.texture_indices_comp = mesh.texture_indices_comp.empty() ? nullptr : mesh.texture_indices_comp.data();
	a.mesh.material_indices = mesh.material_indices.data();
	a.mesh.materials = mesh.materials.data();
	a.mesh.tex_bias[0] = mesh.tex_bias[0]; a.mesh.tex_bias[1] = mesh.tex_bias[1];
	a.mesh.tex_scale[0] = mesh.tex_scale[0]; a.mesh.tex_scale[1] = mesh.tex_scale[1];
	a.mesh.texture_data = mesh.texture_indices_comp.empty() ? nullptr : mesh.texture_data.data();
	a.textures = texture_views.empty() ? nullptr : texture_views.data(); a.num_textures = uint32(texture_views.size());
	a.dir_lights = dir_lights.empty() ? nullptr : dir_lights.data(); a.dir_lights_count = uint32(dir_lights.size());
	a.glossy_reflectance = glossy_reflectance.data();
	a.camera = override_camera ? *override_camera : (cameras.empty() ? default_camera() : cameras[0]);
	a.samples_dir = data_dir.c_str();
	return a;
}

} // namespace fermat

// ---- C hooks (declared in include/fermat_host.h) --------------------------------------------------------------------------------
extern "C" {
static thread_local std::string g_scene_error;
void* fpt_host_scene_load(const char* filename, const char* data_dir)
{
	fermat::HostScene* s = new fermat::HostScene();
	try { s->load(filename, data_dir); return s; }
	catch (const std::exception& e) { g_scene_error = e.what(); delete s; return nullptr; }
}
const char* fpt_host_scene_last_error() { return g_scene_error.c_str(); }
void fpt_host_scene_free(void* h) { delete static_cast<fermat::HostScene*>(h); }
int fpt_host_scene_arrays(const void* h, const fpt_camera* override_camera, fermat::SceneArrays* out)
{
	if (!h || !out) return 1;
	*out = static_cast<const fermat::HostScene*>(h)->arrays(override_camera);
	return 0;
}
// counts: [0] cameras, [1] dir lights, [2] textures, [3] groups
int fpt_host_scene_counts(const void* h, uint32_t out[4])
{
	const fermat::HostScene* s = static_cast<const fermat::HostScene*>(h);
	if (!s) return 1;
	out[0] = uint32_t(s->cameras.size()); out[1] = uint32_t(s->dir_lights.size()); out[2] = uint32_t(s->mesh.textures.size()); out[3] = uint32_t(s->mesh.group_names.size());
	return 0;
}
const char* fpt_host_scene_texture_name(const void* h, uint32_t i)
{
	const fermat::HostScene* s = static_cast<const fermat::HostScene*>(h);
	return (s && i < s->mesh.textures.size()) ? s->mesh.textures[i].c_str() : nullptr;
}
const char* fpt_host_scene_material_name(const void* h, uint32_t i)
{
	const fermat::HostScene* s = static_cast<const fermat::HostScene*>(h);
	return (s && i < s->mesh.material_names.size()) ? s->mesh.material_names[i].c_str() : nullptr;
}
const char* fpt_host_scene_group_name(const void* h, uint32_t i, int32_t* first_triangle, int32_t* end_triangle)
{
	const fermat::HostScene* s = static_cast<const fermat::HostScene*>(h);
	if (!s || i >= s->mesh.group_names.size()) return nullptr;
	if (first_triangle) *first_triangle = s->mesh.group_offsets[i];
	if (end_triangle) *end_triangle = s->mesh.group_offsets[i + 1];
	return s->mesh.group_names[i].c_str();
}
int fpt_host_load_camera(const char* filename, fpt_camera* out) { return fermat::load_camera_file(filename, *out) ? 0 : 1; }
int fpt_host_write_tga(const char* filename, int width, int height, const unsigned char* pixels, int channels)
{ return fermat::write_tga(filename, width, height, pixels, channels) ? 0 : 1; }
}
