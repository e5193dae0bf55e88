#pragma once   // TEST-ONLY stub (see README.md): declarations only, written from the call sites in
               // /root/reference/src/dab_module.cpp and src/render_radio_block.cpp; implements nothing, never linked
#include <cinttypes>
#include <cstddef>
#include <cstdint>
struct ImVec2 {
    float x, y;
    ImVec2() : x(0.0f), y(0.0f) {}
    ImVec2(float _x, float _y) : x(_x), y(_y) {}
};
typedef void *ImTextureID;
typedef int ImGuiTableFlags;
typedef int ImGuiTableColumnFlags;
typedef int ImGuiSliderFlags;
enum { ImGuiSliderFlags_AlwaysClamp = 1 << 4 };
enum { ImGuiTableFlags_Resizable = 1 << 0, ImGuiTableFlags_RowBg = 1 << 6, ImGuiTableFlags_Borders = 15 << 7,
       ImGuiTableFlags_SizingStretchSame = 4 << 13 };
enum { ImGuiTableColumnFlags_WidthStretch = 1 << 3 };
namespace ImGui {
bool BeginTabBar(const char *id);
void EndTabBar();
bool BeginTabItem(const char *label);
void EndTabItem();
bool Button(const char *label);
void Separator();
void SameLine(float offset_from_start_x = 0.0f);
void Text(const char *fmt, ...);
void TextWrapped(const char *fmt, ...);
void SetTooltip(const char *fmt, ...);
bool RadioButton(const char *label, bool active);
bool Checkbox(const char *label, bool *v);
bool SliderFloat(const char *label, float *v, float v_min, float v_max, const char *format = "%.3f", ImGuiSliderFlags flags = 0);
bool SliderFloat2(const char *label, float v[2], float v_min, float v_max, const char *format = "%.3f", ImGuiSliderFlags flags = 0);
bool SliderInt(const char *label, int *v, int v_min, int v_max, const char *format = "%d", ImGuiSliderFlags flags = 0);
bool BeginCombo(const char *label, const char *preview_value);
void EndCombo();
void PushID(int int_id);
void PopID();
bool Selectable(const char *label, bool selected = false);
ImVec2 GetContentRegionAvail();
ImVec2 CalcTextSize(const char *text);
ImVec2 GetWindowContentRegionMin();
ImVec2 GetWindowContentRegionMax();
void Image(ImTextureID id, const ImVec2 &size);
bool IsItemHovered();
bool BeginTable(const char *id, int columns, ImGuiTableFlags flags = 0);
void EndTable();
void TableNextRow();
bool TableSetColumnIndex(int column_n);
void TableSetupColumn(const char *label, ImGuiTableColumnFlags flags = 0);
void TableHeadersRow();
void PushItemWidth(float item_width);
void PopItemWidth();
void SetNextItemWidth(float item_width);
}
