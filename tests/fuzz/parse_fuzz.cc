// Host-only build of the library's file parsers (vbt_amd/csrc/container_parse.h) for the sanitizer run of
// tests/test_parser_fuzz.py:  g++ -fsanitize=address,undefined parse_fuzz.cc -o parse_fuzz
//   parse_fuzz container <dir>          every file of <dir> through read_container (+ validate_container)
//   parse_fuzz plan <good plan> <dir>   every file of <dir> through parse_plan_file against the shape of <good plan>
// One line per file: "ok <name>" or "refused <name>: <why>".  A crash / sanitizer report is the failure the test looks for.
#include <dirent.h>

#include <algorithm>
#include <string>

#include "../../vbt_amd/csrc/container_parse.h"

static std::vector<std::string> files_of(const char* dir) {
  std::vector<std::string> out;
  DIR* d = opendir(dir);
  if (!d) return out;
  while (dirent* e = readdir(d))
    if (e->d_name[0] != '.') out.push_back(std::string(dir) + "/" + e->d_name);
  closedir(d);
  std::sort(out.begin(), out.end());
  return out;
}

// the shape a plan file implies for itself: per group `chosen + 1` alternatives with the file's own steps / families
static bool shape_of(const char* path, vbt::PlanShape* shape) {
  FILE* f = fopen(path, "r");
  if (!f) return false;
  char head[32];
  int ng = 0;
  if (fscanf(f, "%31s %d", head, &ng) != 2 || strcmp(head, "VBTPLAN2") != 0 || ng < 1 || ng > 10000) { fclose(f); return false; }
  for (int g = 0; g < ng; g++) {
    int ch = 0, ns = 0;
    if (fscanf(f, "%d %d", &ch, &ns) != 2 || ch < 0 || ch > 64 || ns < 1 || ns > 64) { fclose(f); return false; }
    std::vector<vbt::PlanStepShape> steps;
    for (int i = 0; i < ns; i++) {
      char tok[96];
      if (fscanf(f, "%95s", tok) != 1) { fclose(f); return false; }
      char* colon = strrchr(tok, ':');
      if (!colon) { fclose(f); return false; }
      *colon = 0;
      vbt::PlanStepShape s;
      s.family = tok;
      s.variants = {-1, 0, 1, 2, 3, 4, 5, 6, 9, 11, 17, 25, 100, 101, 102, 103, 104, 106, 201, 202, 203, 204, 206};
      steps.push_back(s);
    }
    shape->groups.push_back(std::vector<std::vector<vbt::PlanStepShape>>((size_t)ch + 1, steps));
  }
  fclose(f);
  return true;
}

int main(int argc, char** argv) {
  if (argc == 3 && strcmp(argv[1], "container") == 0) {
    for (const std::string& p : files_of(argv[2])) {
      vbt::ContainerData c;
      std::string err;
      if (vbt::read_container(p.c_str(), &c, &err)) printf("ok %s\n", p.c_str());
      else printf("refused %s: %s\n", p.c_str(), err.c_str());
    }
    return 0;
  }
  if (argc == 4 && strcmp(argv[1], "plan") == 0) {
    vbt::PlanShape shape;
    if (!shape_of(argv[2], &shape)) { fprintf(stderr, "cannot read the reference plan %s\n", argv[2]); return 2; }
    for (const std::string& p : files_of(argv[3])) {
      std::vector<vbt::PlanChoice> sel;
      std::string note;
      if (vbt::parse_plan_file(p.c_str(), shape, &sel, &note)) printf("ok %s\n", p.c_str());
      else printf("refused %s: %s\n", p.c_str(), note.c_str());
    }
    return 0;
  }
  fprintf(stderr, "usage: parse_fuzz container <dir> | parse_fuzz plan <good plan> <dir>\n");
  return 2;
}
